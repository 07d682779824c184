// libjqc_hip.so -- C-ABI host layer + class-independent device kernels (gfx950).
// Declarations and the reference interfaces each entry point replaces: include/jqc_hip.h.
//
// Class-specialised J/K kernels are compiled from joltqc_amd/csrc/kernels/*.hip through hiprtc into
// per-class gfx950 code objects cached on disk (the MI355X analogue of the reference's
// CuPy RawModule + cubin cache, /root/reference/jqc/backend/jk_1q1t.py:117-148), then loaded with
// the HIP module API and launched on caller-provided device pointers.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

#include "../../include/jqc_hip.h"

namespace {

thread_local std::string g_err;
std::mutex g_mu;
std::string g_src_dir, g_cache_dir;
int g_tsw[5] = {8, 4, 4, 2, 1};     // shells per tile edge by angular momentum (jqc_set_tile_widths)
// per-share partial sums of the split VV10 inner loop (jqc_vv10), one buffer PER STREAM: calls on one stream are ordered by the
// stream, calls on different streams (or host threads) no longer share a buffer; released by jqc_release_scratch
std::map<void*, std::pair<double*, size_t>> g_vv10_scratch;
std::string g_src_tag = "nosrc";   // FNV-1a of every kernel source: stale code objects are never reused
std::string g_pair_tag = "nosrc";  // pair-based J kernels (pair_vj.hip + common headers)
std::string g_grad_tag = "nosrc";  // same for the gradient kernels (jk_grad.hip + the headers it includes), kept apart so that
                                   // work on them does not invalidate the verified J/K code objects

int fail(int code, const char* fmt, ...)
{
    char buf[4096];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_OK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return fail(-2, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct Kernel {
    std::string key;
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    int li = 0, lj = 0, lk = 0, ll = 0, fp32 = 0, algo = 0, nroots = 1;
    int per_wg = 0;         // gradient kernels: quartets per workgroup pass (0: one quartet per lane)
};
// a deque: handles stay valid and references to entries are not moved when another thread registers a kernel; every
// access to the container itself (size, lookup, push_back) happens under g_mu, the launchers copy the few fields they need
std::deque<Kernel> g_kernels;
std::map<std::string, int> g_by_key;

struct KernelView { hipFunction_t fn; int fp32, algo, nroots, per_wg; };
bool kernel_view(int handle, KernelView& v)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (handle < 0 || handle >= (int)g_kernels.size() || !g_kernels[handle].fn) return false;
    const Kernel& k = g_kernels[handle];
    v = KernelView{k.fn, k.fp32, k.algo, k.nroots, k.per_wg};
    return true;
}

// the fixed hiprtc option set (part of the source tag: a code object built with other options is another build)
const char* const kHiprtcOpts[] = {"--offload-arch=gfx950", "-O3", "-munsafe-fp-atomics", "-ffp-contract=fast"};

// device copies of the Rys tables
double* g_rys64 = nullptr;
float* g_rys32 = nullptr;
std::vector<double> g_rys_host;

bool file_exists(const std::string& p)
{
    struct stat st;
    return stat(p.c_str(), &st) == 0 && st.st_size > 0;
}

std::string read_file(const std::string& p)
{
    std::ifstream f(p, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

// hiprtc: source file + defines -> code object in memory
int compile_code(const std::string& src_name, const std::vector<std::string>& defs, std::string& code)
{
    const std::string path = g_src_dir + "/" + src_name;
    const std::string src = read_file(path);
    if (src.empty()) return fail(-3, "kernel source %s not found (jqc_set_kernel_dirs?)", path.c_str());
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), src_name.c_str(), 0, nullptr, nullptr) != HIPRTC_SUCCESS)
        return fail(-3, "hiprtcCreateProgram failed");
    std::vector<std::string> opts(std::begin(kHiprtcOpts), std::end(kHiprtcOpts));
    opts.push_back("-I" + g_src_dir);
    for (auto& d : defs) opts.push_back(d);
    if (const char* extra = getenv("JQC_EXTRA_DEFS")) {      // tuning experiments: e.g. "-DMINW=3 -DECAP=32"
        std::stringstream ss(extra);
        std::string tok;
        while (ss >> tok) opts.push_back(tok);
    }
    std::vector<const char*> copts;
    for (auto& o : opts) copts.push_back(o.c_str());
    hiprtcResult r = hiprtcCompileProgram(prog, (int)copts.size(), copts.data());
    if (r != HIPRTC_SUCCESS) {
        size_t n = 0;
        hiprtcGetProgramLogSize(prog, &n);
        std::string log(n + 1, '\0');
        hiprtcGetProgramLog(prog, &log[0]);
        hiprtcDestroyProgram(&prog);
        return fail(-3, "hiprtc compile of %s failed:\n%s", src_name.c_str(), log.c_str());
    }
    size_t n = 0;
    hiprtcGetCodeSize(prog, &n);
    code.assign(n, '\0');
    hiprtcGetCode(prog, &code[0]);
    hiprtcDestroyProgram(&prog);
    return 0;
}

// code object -> disk (atomic rename so concurrent ranks are safe)
int write_code(const std::string& code, const std::string& out)
{
    char tmp[64];
    snprintf(tmp, sizeof tmp, ".tmp.%d", (int)getpid());
    const std::string t = out + tmp;
    {
        std::ofstream f(t, std::ios::binary);
        f.write(code.data(), (std::streamsize)code.size());
        if (!f) return fail(-3, "cannot write %s", t.c_str());
    }
    if (rename(t.c_str(), out.c_str()) != 0) return fail(-3, "cannot rename %s", t.c_str());
    return 0;
}

int compile_to(const std::string& src_name, const std::vector<std::string>& defs, const std::string& out)
{
    std::string code;
    int rc = compile_code(src_name, defs, code);
    return rc ? rc : write_code(code, out);
}

// bytes of scratch (register spill space) per lane of the first kernel of a code object: the msgpack unsigned that follows
// the ".private_segment_fixed_size" key of its metadata note
long scratch_bytes(const std::string& code)
{
    static const char key[] = ".private_segment_fixed_size";
    const size_t at = code.find(key);
    if (at == std::string::npos) return -1;
    const unsigned char* p = (const unsigned char*)code.data() + at + sizeof(key) - 1;
    const size_t left = code.size() - (at + sizeof(key) - 1);
    if (left < 5) return -1;
    if (p[0] < 0x80) return p[0];
    if (p[0] == 0xcc) return p[1];
    if (p[0] == 0xcd) return (p[1] << 8) | p[2];
    if (p[0] == 0xce) return ((long)p[1] << 24) | (p[2] << 16) | (p[3] << 8) | p[4];
    return -1;
}

int load_kernel(const std::string& hsaco, const char* entry, Kernel& k)
{
    const std::string code = read_file(hsaco);
    if (code.empty()) return fail(-3, "code object %s missing", hsaco.c_str());
    HIP_OK(hipModuleLoadData(&k.mod, code.data()));
    HIP_OK(hipModuleGetFunction(&k.fn, k.mod, entry));
    return 0;
}

inline const double* rys_cheb64(int n) { return g_rys64 + (long)g_rys_host[2 * n - 1]; }
inline const double* rys_large64(int n) { return g_rys64 + (long)g_rys_host[2 * n]; }
inline const float* rys_cheb32(int n) { return g_rys32 + (long)g_rys_host[2 * n - 1]; }
inline const float* rys_large32(int n) { return g_rys32 + (long)g_rys_host[2 * n]; }

// ------------------------------------------------------------------------------------------------
// Device kernels that do not depend on the angular-momentum class
// ------------------------------------------------------------------------------------------------

// One block = 16 ij pairs x 16 kl pairs of one screen task.  Survivors are compacted with wave
// ballots (wave64) + one atomic per block and list.
__global__ void __launch_bounds__(256)
screen_kernel(const int* __restrict__ tasks, const int ntasks, const unsigned* __restrict__ pair_sh,
              const float* __restrict__ pair_q, const float* __restrict__ log_dm, const int nbas, const int do_j,
              const int do_k, const float cut32, const float cut64, const float log_max_dm,
              ushort4* __restrict__ queue, const long long* __restrict__ region, unsigned* __restrict__ counters)
{
    __shared__ int s_task[8];
    __shared__ unsigned s_cnt[2][4];
    __shared__ unsigned s_base[2];
    const int tid = threadIdx.x;
    if (tid == 0) {
        // binary search: last task with blk0 <= blockIdx.x
        int lo = 0, hi = ntasks - 1;
        const int b = blockIdx.x;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tasks[mid * 8 + 5] <= b) lo = mid; else hi = mid - 1;
        }
        for (int n = 0; n < 8; n++) s_task[n] = tasks[lo * 8 + n];
    }
    __syncthreads();
    const int ij0 = s_task[0], nij = s_task[1], kl0 = s_task[2], nkl = s_task[3], cls = s_task[4];
    const int lb = blockIdx.x - s_task[5];
    const int nbk = (nkl + 15) >> 4;
    const int bij = lb / nbk, bkl = lb - bij * nbk;
    // pair lists are sorted by q descending: the first pair of each 16-strip carries the strip maximum
    const float qmax = pair_q[ij0 + bij * 16] + pair_q[kl0 + bkl * 16];
    if (qmax + log_max_dm <= cut32) return;

    const int ij = bij * 16 + (tid >> 4), kl = bkl * 16 + (tid & 15);
    int keep = 0;  // 1: fp64 list, 2: fp32 list
    ushort4 sq = {0, 0, 0, 0};
    if (ij < nij && kl < nkl) {
        const unsigned pij = pair_sh[ij0 + ij], pkl = pair_sh[kl0 + kl];
        const int ish = pij >> 16, jsh = pij & 0xffff, ksh = pkl >> 16, lsh = pkl & 0xffff;
        if (ish * nbas + jsh >= ksh * nbas + lsh) {
            const float q = pair_q[ij0 + ij] + pair_q[kl0 + kl];
            float d = -36.8f;
            if (do_k) {
                d = fmaxf(d, log_dm[ish * nbas + ksh]);
                d = fmaxf(d, log_dm[jsh * nbas + ksh]);
                d = fmaxf(d, log_dm[ish * nbas + lsh]);
                d = fmaxf(d, log_dm[jsh * nbas + lsh]);
            }
            if (do_j) {
                d = fmaxf(d, log_dm[ish * nbas + jsh]);
                d = fmaxf(d, log_dm[ksh * nbas + lsh]);
            }
            const float dq = q + d;
            if (dq > cut32) keep = (dq > cut64) ? 1 : 2;
            sq.x = ish; sq.y = jsh; sq.z = ksh; sq.w = lsh;
        }
    }
    const int wave = tid >> 6, lane = tid & 63;
    const unsigned long long m64 = __ballot(keep == 1), m32 = __ballot(keep == 2);
    if (lane == 0) {
        s_cnt[0][wave] = __popcll(m64);
        s_cnt[1][wave] = __popcll(m32);
    }
    __syncthreads();
    if (tid < 2) {
        const unsigned tot = s_cnt[tid][0] + s_cnt[tid][1] + s_cnt[tid][2] + s_cnt[tid][3];
        s_base[tid] = tot ? atomicAdd(&counters[cls * 2 + tid], tot) : 0u;
    }
    __syncthreads();
    if (keep) {
        const int w = keep - 1;
        unsigned off = s_base[w];
        for (int x = 0; x < wave; x++) off += s_cnt[w][x];
        const unsigned long long m = w ? m32 : m64;
        off += __popcll(m & ((1ull << lane) - 1ull));
        const long long pos = w ? region[cls * 2 + 1] - 1 - (long long)off : region[cls * 2] + (long long)off;
        queue[pos] = sq;
    }
}

// Primitive-pair prefactors of every shell pair of every tile pair (once per geometry): for tile pair t with widths
// (wi, wj) the block out[pp_off[t]*27 ...] holds, for shell pair (a, b) and primitive pair (p1, p2),
//   {c_a c_b exp(-a1 a2 / (a1+a2) |Ra-Rb|^2), 1/(a1+a2), a1+a2}        (reference 1q1t.cu:146-171 caches the same K_ab)
// at ((a*wj + b)*9 + p1*3 + p2)*3.  Slots of primitives a shell does not have are never read by the J/K kernels.
__global__ void __launch_bounds__(64)
pair_table_kernel(const double* __restrict__ basis, const unsigned* __restrict__ tpair_sh,
                  const unsigned* __restrict__ tpair_wij, const unsigned* __restrict__ pp_off, double* __restrict__ out)
{
    const int t = blockIdx.x;
    const unsigned p = tpair_sh[t], w = tpair_wij[t];
    const int ish0 = p >> 16, jsh0 = p & 0xffff, wi = w >> 16, wj = w & 0xffff;
    double* __restrict__ dst = out + (size_t)pp_off[t] * 27;
    for (int n = threadIdx.x; n < wi * wj * 9; n += 64) {
        const int pr = n / 9, pp = n - pr * 9, p1 = pp / 3, p2 = pp - p1 * 3;
        const double* s1 = basis + (size_t)(ish0 + pr / wj) * 12;
        const double* s2 = basis + (size_t)(jsh0 + pr % wj) * 12;
        const double dx = s2[0] - s1[0], dy = s2[1] - s1[1], dz = s2[2] - s1[2];
        const double a1 = s1[5 + 2 * p1], a2 = s2[5 + 2 * p2];
        const double asum = a1 + a2;
        const bool have = p1 < (int)s1[10] && p2 < (int)s2[10];
        const double inv = have ? 1.0 / asum : 0.0;
        dst[n * 3 + 0] = have ? s1[4 + 2 * p1] * s2[4 + 2 * p2] * exp(-a1 * a2 * inv * (dx * dx + dy * dy + dz * dz)) : 0.0;
        dst[n * 3 + 1] = inv;
        dst[n * 3 + 2] = have ? asum : 0.0;
    }
}

__global__ void __launch_bounds__(256)
shell_block_max_kernel(const double* __restrict__ mat, const int n_dm, const int nao, const int* __restrict__ ao_loc,
                       const int nbas, float* __restrict__ out)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nbas * nbas) return;
    const int I = idx / nbas, J = idx - I * nbas;
    const int r0 = ao_loc[I], r1 = ao_loc[I + 1], c0 = ao_loc[J], c1 = ao_loc[J + 1];
    double m = 0;
    for (int b = 0; b < n_dm; b++)
        for (int r = r0; r < r1; r++)
            for (int c = c0; c < c1; c++) m = fmax(m, fabs(mat[((size_t)b * nao + r) * nao + c]));
    out[idx] = (float)m;
}

// E coefficients of the pair-based J kernel (pair_vj.hip): per ket pair (k >= l)
//   E[cx,cy,cz] = (2 - delta_kl) sum_{k comp, l comp} D[l, k] Hx(kx,lx;cx) Hy(ky,ly;cy) Hz(kz,lz;cz),
//   H(k,l;c) = C(l, c-k) (R_k - R_l)^(l-c+k)    (ket horizontal recurrence g(k,l+1) = g(k+1,l) - (R_l - R_k) g(k,l) unrolled),
// triples ordered by total cx+cy+cz = lk..lk+ll ascending, cx descending, cy descending.  One lane per ket pair.
__global__ void __launch_bounds__(64)
pair_ket_density_kernel(const double* __restrict__ basis, const double* __restrict__ dm, const int nao,
                        const unsigned* __restrict__ ket_pairs, const int n_ket, const int lk, const int ll,
                        double* __restrict__ E, float* __restrict__ ld)
{
    const int kt = blockIdx.x * 64 + threadIdx.x;
    if (kt >= n_ket) return;
    const unsigned p = ket_pairs[kt];
    const int ksh = p >> 16, lsh = p & 0xffff;
    const double* bk = basis + (size_t)ksh * 12;
    const double* bl = basis + (size_t)lsh * 12;
    const int k0 = (int)bk[3], l0 = (int)bl[3];
    const int lkl = lk + ll;
    // H[ax][k][l][c]
    double H[3][5][5][9];
    for (int ax = 0; ax < 3; ax++) {
        const double r = bk[ax] - bl[ax];
        for (int k = 0; k <= lk; k++)
            for (int l = 0; l <= ll; l++)
                for (int c = 0; c <= lkl; c++) {
                    const int m = c - k;
                    double v = 0;
                    if (m >= 0 && m <= l) {
                        double binom = 1;
                        for (int q = 0; q < m; q++) binom = binom * (l - q) / (q + 1);
                        v = binom;
                        for (int q = 0; q < l - m; q++) v *= r;
                    }
                    H[ax][k][l][c] = v;
                }
    }
    const double f = ksh == lsh ? 1.0 : 2.0;
    const int nfk = (lk + 1) * (lk + 2) / 2, nfl = (ll + 1) * (ll + 2) / 2;
    int ntrip = 0;
    for (int t = lk; t <= lkl; t++) ntrip += (t + 1) * (t + 2) / 2;
    double* out = E + (size_t)kt * ntrip;
    double dmax = 0;
    int n = 0;
    for (int tot = lk; tot <= lkl; tot++)
        for (int cx = tot; cx >= 0; cx--)
            for (int cy = tot - cx; cy >= 0; cy--, n++) {
                const int cz = tot - cx - cy;
                double s = 0;
                int kc = 0;
                for (int kx = lk; kx >= 0; kx--)
                    for (int ky = lk - kx; ky >= 0; ky--, kc++) {
                        const int kz = lk - kx - ky;
                        int lc = 0;
                        for (int lx = ll; lx >= 0; lx--)
                            for (int ly = ll - lx; ly >= 0; ly--, lc++) {
                                const int lz = ll - lx - ly;
                                const double d = dm[(size_t)(l0 + lc) * nao + k0 + kc];
                                if (n == 0) dmax = fmax(dmax, fabs(d));
                                s += d * H[0][kx][lx][cx] * H[1][ky][ly][cy] * H[2][kz][lz][cz];
                            }
                    }
                out[n] = f * s;
            }
    (void)nfk; (void)nfl;
    ld[kt] = (float)log(dmax + 1e-300);
}

// One-electron integrals S, T, V of every shell pair (i >= j) in the internal Cartesian basis (the
// last CPU / libcint step before the Fock build; the reference takes them from PySCF).  One lane per shell pair, run-time
// angular momenta (a once-per-geometry kernel).  Obara-Saika recurrences for overlap and kinetic energy; nuclear attraction
// by Rys quadrature with the same tables as the ERIs: V_ab = -sum_C Z_C 2 pi / p K_ab sum_r w_r Ix Iy Iz,
// I(i+1,0) = (PA - t_r^2 PC) I(i,0) + i (1 - t_r^2) / (2p) I(i-1,0), roots at x = p |PC|^2.
// Cartesian order and normalisation are those of the packed shell table (s, p coefficients carry sqrt((2l+1)/4pi)).
__global__ void __launch_bounds__(64)
int1e_kernel(const double* __restrict__ basis, const int* __restrict__ ao_loc, const unsigned* __restrict__ pairs, const int npairs,
             const double* __restrict__ atoms, const int natm, const int nao, const double* __restrict__ rys, const double* __restrict__ rys_hdr,
             double* __restrict__ S, double* __restrict__ T, double* __restrict__ V)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= npairs) return;
    const unsigned p = pairs[t];
    const int ish = p >> 16, jsh = p & 0xffff;
    const double* bi = basis + (size_t)ish * 12;
    const double* bj = basis + (size_t)jsh * 12;
    const int li = (int)bi[11], lj = (int)bj[11], npi = (int)bi[10], npj = (int)bj[10];
    const int nfi = (li + 1) * (li + 2) / 2, nfj = (lj + 1) * (lj + 2) / 2;
    const int i0 = ao_loc[ish], j0 = ao_loc[jsh];
    const double A[3] = {bi[0], bi[1], bi[2]}, B[3] = {bj[0], bj[1], bj[2]};
    const double AB[3] = {A[0] - B[0], A[1] - B[1], A[2] - B[2]};
    const double r2 = AB[0] * AB[0] + AB[1] * AB[1] + AB[2] * AB[2];
    double s_blk[15 * 15], t_blk[15 * 15], v_blk[15 * 15];
    for (int n = 0; n < nfi * nfj; n++) s_blk[n] = t_blk[n] = v_blk[n] = 0;
    const int nroots = (li + lj) / 2 + 1;
    const double* cheb = rys + (long)rys_hdr[2 * nroots - 1];
    const double* large = rys + (long)rys_hdr[2 * nroots];
    for (int ip = 0; ip < npi; ip++)
    for (int jp = 0; jp < npj; jp++) {
        const double a = bi[5 + 2 * ip], b = bj[5 + 2 * jp], cc = bi[4 + 2 * ip] * bj[4 + 2 * jp];
        const double pp = a + b, ip2 = 0.5 / pp;
        const double K = cc * exp(-a * b / pp * r2);
        double P[3], PA[3];
        for (int x = 0; x < 3; x++) { P[x] = (a * A[x] + b * B[x]) / pp; PA[x] = P[x] - A[x]; }
        // ---- overlap 1-D integrals s[x][i][j], i <= li, j <= lj + 2 (kinetic needs j + 2)
        double s1[3][5][7];
        const double s00 = sqrt(3.14159265358979323846 / pp);
        for (int x = 0; x < 3; x++) {
            double col[12];                       // (m, 0), m <= li + lj + 2
            col[0] = s00;
            col[1] = PA[x] * s00;
            for (int m = 1; m < li + lj + 2; m++) col[m + 1] = PA[x] * col[m] + m * ip2 * col[m - 1];
            // (i, j+1) = (i+1, j) + (A - B) (i, j)
            double w[12];
            for (int m = 0; m <= li + lj + 2; m++) w[m] = col[m];
            for (int j = 0; j <= lj + 2; j++) {
                for (int i = 0; i <= li; i++) s1[x][i][j] = w[i];
                for (int m = 0; m < li + lj + 2 - j; m++) w[m] = w[m + 1] + AB[x] * w[m];
            }
        }
        // kinetic 1-D: t(i,j) = -2 b^2 s(i,j+2) + b (2j+1) s(i,j) - j(j-1)/2 s(i,j-2)
        double t1[3][5][5];
        for (int x = 0; x < 3; x++)
            for (int i = 0; i <= li; i++)
                for (int j = 0; j <= lj; j++)
                    t1[x][i][j] = -2.0 * b * b * s1[x][i][j + 2] + b * (2 * j + 1) * s1[x][i][j] -
                                  (j >= 2 ? 0.5 * j * (j - 1) * s1[x][i][j - 2] : 0.0);
        int ci = 0;
        for (int ix = li; ix >= 0; ix--)
        for (int iy = li - ix; iy >= 0; iy--, ci++) {
            const int iz = li - ix - iy;
            int cj = 0;
            for (int jx = lj; jx >= 0; jx--)
            for (int jy = lj - jx; jy >= 0; jy--, cj++) {
                const int jz = lj - jx - jy;
                const double sx = s1[0][ix][jx], sy = s1[1][iy][jy], sz = s1[2][iz][jz];
                s_blk[ci * nfj + cj] += K * sx * sy * sz;
                t_blk[ci * nfj + cj] += K * (t1[0][ix][jx] * sy * sz + sx * t1[1][iy][jy] * sz + sx * sy * t1[2][iz][jz]);
            }
        }
        // ---- nuclear attraction
        for (int c = 0; c < natm; c++) {
            const double Z = atoms[4 * c + 3];
            const double PC[3] = {P[0] - atoms[4 * c], P[1] - atoms[4 * c + 1], P[2] - atoms[4 * c + 2]};
            const double x = pp * (PC[0] * PC[0] + PC[1] * PC[1] + PC[2] * PC[2]);
            double rw[18];
            // roots (t^2) and weights at x (same tables / branches as rys_roots in kernels/jk_common.h)
            if (x >= 5 * nroots + 35) {
                const double isx = 1.0 / sqrt(x), ix2 = isx * isx;
                for (int r = 0; r < nroots; r++) { rw[2 * r] = large[2 * r] * ix2; rw[2 * r + 1] = large[2 * r + 1] * isx; }
            } else {
                const int it = (int)(x * 0.4);
                const double u = (x - 2.5 * it) * 0.8 - 1.0, u2 = u + u;
                const double* cf = cheb + (size_t)it * nroots * 28;
                for (int r = 0; r < nroots; r++, cf += 28) {
                    double br1 = 0, br2 = 0, bw1 = 0, bw2 = 0;
                    for (int k = 13; k >= 1; k--) {
                        double q = cf[2 * k] + u2 * br1 - br2; br2 = br1; br1 = q;
                        q = cf[2 * k + 1] + u2 * bw1 - bw2; bw2 = bw1; bw1 = q;
                    }
                    rw[2 * r] = cf[0] + u * br1 - br2;
                    rw[2 * r + 1] = cf[1] + u * bw1 - bw2;
                }
            }
            const double pref = -Z * K * 2.0 * 3.14159265358979323846 / pp;
            for (int r = 0; r < nroots; r++) {
                // the tables hold weights of int_0^1 with the ERI convention sum_r w_r = F_0(x) (tests/test_rys.py)
                const double t2 = rw[2 * r], wt = rw[2 * r + 1];
                double g[3][5][5];
                for (int xx = 0; xx < 3; xx++) {
                    double col[9];
                    const double c0 = PA[xx] - t2 * PC[xx], b10 = ip2 * (1.0 - t2);
                    col[0] = 1.0;
                    col[1] = c0;
                    for (int m = 1; m < li + lj; m++) col[m + 1] = c0 * col[m] + m * b10 * col[m - 1];
                    double w[9];
                    for (int m = 0; m <= li + lj; m++) w[m] = col[m];
                    for (int j = 0; j <= lj; j++) {
                        for (int i = 0; i <= li; i++) g[xx][i][j] = w[i];
                        for (int m = 0; m < li + lj - j; m++) w[m] = w[m + 1] + AB[xx] * w[m];
                    }
                }
                int ci2 = 0;
                for (int ix = li; ix >= 0; ix--)
                for (int iy = li - ix; iy >= 0; iy--, ci2++) {
                    const int iz = li - ix - iy;
                    int cj = 0;
                    for (int jx = lj; jx >= 0; jx--)
                    for (int jy = lj - jx; jy >= 0; jy--, cj++) {
                        const int jz = lj - jx - jy;
                        v_blk[ci2 * nfj + cj] += pref * wt * g[0][ix][jx] * g[1][iy][jy] * g[2][iz][jz];
                    }
                }
            }
        }
    }
    for (int ci = 0; ci < nfi; ci++)
        for (int cj = 0; cj < nfj; cj++) {
            const size_t ij = (size_t)(i0 + ci) * nao + j0 + cj, ji = (size_t)(j0 + cj) * nao + i0 + ci;
            S[ij] = S[ji] = s_blk[ci * nfj + cj];
            T[ij] = T[ji] = t_blk[ci * nfj + cj];
            V[ij] = V[ji] = v_blk[ci * nfj + cj];
        }
}

#include "dft_kernels.inc"
#include "ecp_kernels.inc"

}  // namespace

// ------------------------------------------------------------------------------------------------
extern "C" {

const char* jqc_last_error(void) { return g_err.c_str(); }
const char* jqc_version(void) { return "joltqc_amd 0.1 (gfx950)"; }
const char* jqc_source_tag(void) { return g_src_tag.c_str(); }
const char* jqc_grad_source_tag(void) { return g_grad_tag.c_str(); }
const char* jqc_pair_source_tag(void) { return g_pair_tag.c_str(); }

int jqc_set_kernel_dirs(const char* src_dir, const char* cache_dir)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_src_dir = src_dir ? src_dir : "";
    g_cache_dir = cache_dir ? cache_dir : "";
    if (!g_cache_dir.empty()) mkdir(g_cache_dir.c_str(), 0755);
    // One tag per kernel family = FNV-1a of the family's own source + the shared headers + everything that decides the build
    // (extra definitions, build policy of this file, hiprtc version and options): an edit of jk_tile.hip does not invalidate
    // the pair-J or gradient code objects, and vice versa.  The verified-build manifest is keyed on the J/K tag.
    auto mix = [](unsigned long long h, const std::string& txt) {
        for (unsigned char ch : txt) { h ^= ch; h *= 1099511628211ull; }
        return h;
    };
    unsigned long long h0 = 1469598103934665603ull;
    for (const char* f : {"jk_common.h", "jk_axis.h"}) h0 = mix(h0, read_file(g_src_dir + "/" + f));
    if (const char* extra = getenv("JQC_EXTRA_DEFS")) h0 = mix(h0, extra);
    // (bump when the build logic of jqc_gen_jk_kernel changes: MINW rebuild loop, ECAP codes, KARG_RELOAD choice, variant bits)
    h0 = mix(h0, "build-policy-r6b:karg-reload-iff-scratch,ored,paroot,ndm2,family-tags,mixed,rsplit,quad,qchunk,hb,hej,kw");
    // compiler version and option set: register allocation decides which builds pass the gates, so code
    // objects of another hiprtc are other builds -- not reused from the cache, not covered by the verified manifest
    {
        int major = 0, minor = 0;
        (void)hiprtcVersion(&major, &minor);
        char ver[64];
        snprintf(ver, sizeof ver, "hiprtc-%d.%d", major, minor);
        h0 = mix(h0, ver);
        for (const char* o : kHiprtcOpts) h0 = mix(h0, o);
    }
    auto tag_of = [&](std::initializer_list<const char*> files) {
        unsigned long long h = h0;
        for (const char* f : files) h = mix(h, read_file(g_src_dir + "/" + f));
        char tag[32];
        snprintf(tag, sizeof tag, "%010llx", h & 0xffffffffffull);
        return std::string(tag);
    };
    g_src_tag = tag_of({"jk_1q1t.hip", "jk_tile.hip", "schwarz.hip"});
    g_pair_tag = tag_of({"pair_vj.hip"});
    g_grad_tag = tag_of({"jk_grad.hip"});
    return 0;
}

int jqc_set_rys_tables(const double* blob, size_t n)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_rys_host.assign(blob, blob + n);
    std::vector<float> f(n);
    for (size_t i = 0; i < n; i++) f[i] = (float)blob[i];
    if (g_rys64) { (void)hipFree(g_rys64); g_rys64 = nullptr; }
    if (g_rys32) { (void)hipFree(g_rys32); g_rys32 = nullptr; }
    HIP_OK(hipMalloc((void**)&g_rys64, n * sizeof(double)));
    HIP_OK(hipMalloc((void**)&g_rys32, n * sizeof(float)));
    HIP_OK(hipMemcpy(g_rys64, blob, n * sizeof(double), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(g_rys32, f.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

int jqc_set_tile_widths(const int* widths5)
{
    std::lock_guard<std::mutex> lk_(g_mu);
    for (int l = 0; l < 5; l++) {
        if (widths5[l] < 1 || widths5[l] > 16) return fail(-1, "tile width of l=%d out of range: %d", l, widths5[l]);
        g_tsw[l] = widths5[l];
    }
    return 0;
}

int jqc_gen_jk_kernel(int li, int lj, int lk, int ll, int do_j, int do_k, int rys_lr, int fp32, int algo_variant,
                      int compile_only)
{
    std::lock_guard<std::mutex> lk_(g_mu);
    // algo_variant: low 4 bits = JQC_ALGO_*, the rest = tuning variant of the tiled kernels (JQC_VARIANT_*)
    const int algo = algo_variant & 0xf;
    const int v_minw = (algo_variant >> 4) & 0xf;
    const int v_rys_l2 = (algo_variant >> 8) & 1, v_st1 = (algo_variant >> 9) & 1, v_wsync = (algo_variant >> 10) & 1, v_cjr = (algo_variant >> 11) & 1;
    const int v_nks = (algo_variant >> 12) & 3;
    const int v_qil = (algo_variant >> 14) & 1, v_cord = (algo_variant >> 15) & 1;
    const int v_ecap = (algo_variant >> 16) & 3;
    const int v_ored = (algo_variant >> 18) & 1, v_paroot = (algo_variant >> 19) & 1, v_ndm2 = (algo_variant >> 20) & 1;
    const int v_mixed = (algo_variant >> 21) & 1, v_rsplit = (algo_variant >> 22) & 3;
    // h form of the row-lane kernels (bit 29); its j-components-per-lane code shares bits 25-26 with the quad chunk code (the two
    // forms exclude each other: quad = lane-per-quartet builds, h form = row-lane builds)
    const int v_hb = (algo_variant >> 29) & 1, v_hej = v_hb ? (algo_variant >> 25) & 3 : 0;
    const int v_quad = (algo_variant >> 24) & 1, v_qnch = v_hb ? 0 : (algo_variant >> 25) & 3, v_qy = v_hb ? 0 : (algo_variant >> 27) & 3;
    const int v_kw = (algo_variant >> 30) & 1;
    if (v_kw && (algo != JQC_ALGO_TILE512 || !((algo_variant >> 18) & 1) || ((algo_variant >> 10) & 1)))
        return fail(-1, "JQC_VARIANT_KW: 512-thread row-lane builds with the owner reduction and workgroup-wide steps");
    if (v_qnch && !v_quad) return fail(-1, "JQC_VARIANT_QCHUNK: quad builds only");
    if (v_hb && ((algo != JQC_ALGO_TILE && algo != JQC_ALGO_TILE512) || !((algo_variant >> 18) & 1) || ((algo_variant >> 11) & 1) || v_quad))
        return fail(-1, "JQC_VARIANT_HB: row-lane builds with the owner reduction (JQC_VARIANT_ORED), without JQC_VARIANT_CJR");
    if (v_quad && (algo != JQC_ALGO_TILE1Q || v_mixed || (li + lj + lk + ll) / 2 + 1 > 4 || (li != 1 && lj != 1 && lk != 1 && ll != 1)))
        return fail(-1, "JQC_VARIANT_QUAD: lane-per-quartet builds of classes with a p shell and at most four Rys roots");
    if (v_mixed && (algo != JQC_ALGO_TILE1Q || fp32 || v_ndm2))
        return fail(-1, "JQC_VARIANT_MIXED: FP64 lane-per-quartet builds with one density matrix per evaluation only");
    if (v_rsplit && algo != JQC_ALGO_TILE && algo != JQC_ALGO_TILE512)
        return fail(-1, "JQC_VARIANT_RSPLIT: row-lane builds only");
    if (li > JQC_LMAX || lj > li || lk > li || ll > lk || li < 0 || lj < 0 || lk < 0 || ll < 0)
        return fail(-1, "unsupported angular class (%d%d|%d%d): need LMAX>=li>=lj, li>=lk>=ll", li, lj, lk, ll);
    if (!do_j && !do_k) return fail(-1, "need do_j or do_k");
    char key[128];
    snprintf(key, sizeof key, "jk%d_%d%d%d%d_j%dk%d_lr%d_%s_t%d.%d.%d.%d.%d", algo_variant, li, lj, lk, ll, do_j, do_k,
             rys_lr, fp32 ? "f32" : "f64", g_tsw[0], g_tsw[1], g_tsw[2], g_tsw[3], g_tsw[4]);
    auto it = g_by_key.find(key);
    if (it != g_by_key.end() && (compile_only || g_kernels[it->second].fn)) return it->second;
    const bool tiled = algo == JQC_ALGO_TILE || algo == JQC_ALGO_TILE1Q || algo == JQC_ALGO_TILE512;
    const char* src = tiled ? "jk_tile.hip" : "jk_1q1t.hip";
    // per-class entry-point name so that rocprofv3 --stats lists every class separately
    char entry[64];
    snprintf(entry, sizeof entry, "%s_%d%d%d%d%s", tiled ? (algo == JQC_ALGO_TILE1Q ? (v_quad ? "jk_quad" : "jk_tile1q") : algo == JQC_ALGO_TILE512 ? "jk_tile512" : "jk_tile") : "jk_1q1t",
             li, lj, lk, ll, fp32 ? "_f32" : (v_mixed ? "_mx" : ""));
    const std::string out = g_cache_dir + "/" + key + "_" + g_src_tag + ".hsaco";
    if (!file_exists(out)) {
        std::vector<std::string> d = {"-DLI=" + std::to_string(li), "-DLJ=" + std::to_string(lj),
                                      "-DLK=" + std::to_string(lk), "-DLL=" + std::to_string(ll),
                                      "-DDO_J=" + std::to_string(do_j), "-DDO_K=" + std::to_string(do_k),
                                      "-DRYS_LR=" + std::to_string(rys_lr), "-DFP32=" + std::to_string(fp32),
                                      "-DTILE_1Q=" + std::to_string(algo == JQC_ALGO_TILE1Q ? 1 : 0),
                                      "-DTBLOCK=" + std::to_string(algo == JQC_ALGO_TILE512 ? 512 : 256),
                                      std::string("-DKNAME=") + entry};
        for (int l = 0; l < 5; l++) d.push_back("-DTSW" + std::to_string(l) + "=" + std::to_string(g_tsw[l]));
        if (v_minw) d.push_back("-DMINW=" + std::to_string(v_minw));
        if (v_rys_l2) d.push_back("-DRYS_LDS_MAX=0");
        if (v_st1) d.push_back("-DST_LDS_MAX=0");
        if (v_wsync) d.push_back("-DWSYNC=1");
        if (v_cjr) d.push_back("-DCJR=1");
        if (v_nks) d.push_back("-DNKS=" + std::to_string(1 << v_nks));
        if (v_qil) d.push_back("-DQIL=1");
        if (v_cord) d.push_back("-DCORD=1");
        if (v_ecap) d.push_back(std::string("-DECAP=") + (v_ecap == 1 ? "32" : v_ecap == 2 ? "16" : "48"));
        if (v_ored) d.push_back("-DORED=1");
        if (v_paroot) d.push_back("-DPAROOT=1");
        if (v_ndm2) d.push_back("-DNDM=2");
        if (v_mixed) d.push_back("-DMIXED=1");
        if (v_rsplit) d.push_back("-DRSPLIT=" + std::to_string(v_rsplit + 1));
        if (v_hb) {
            static const int hej_cap[4] = {1, 2, 3, 6};
            d.push_back("-DHB=1");
            d.push_back("-DHEJ=" + std::to_string(hej_cap[v_hej]));
        }
        if (v_kw) d.push_back("-DKW=1");
        if (v_quad) d.push_back("-DQUAD=1");
        if (v_qnch) {
            static const int nch[4] = {1, 2, 3, 5};
            d.push_back("-DQNCH=" + std::to_string(nch[v_qnch]));
            d.push_back("-DQY=" + std::to_string(v_qy));
        }
        if (tiled) {
            // Builds that spill vector registers to scratch also re-read the staging pointers from the kernarg segment
            // (KARG_RELOAD in jk_tile.hip: ~45 fewer SGPRs spilled to VGPR lanes); builds without scratch keep the
            // pointers in SGPRs, which is 2-20 % faster for the short-iteration classes.  See DESIGN.md section 3.1.
            auto build = [&](std::vector<std::string> dd, std::string& code) -> int {
                dd.push_back("-DKARG_RELOAD=0");
                int rc = compile_code(src, dd, code);
                if (rc) return rc;
                if (scratch_bytes(code) != 0) {
                    dd.back() = "-DKARG_RELOAD=1";
                    rc = compile_code(src, dd, code);
                }
                return rc;
            };
            std::string code;
            int rc = build(d, code);
            if (rc) return rc;
            // The scheme table is tuned on the J+K builds.  The J-only / K-only build of a lane-per-quartet variant has fewer LDS
            // tiles, so the compiler may size its register budget for more workgroups per CU than the variant was tuned for
            // and spill (> 1 KB per lane, 4-5 x slower: profiles/r02_class_profile_j_only_*): if the build spills more than 512 B, the same
            // variant is rebuilt for fewer waves per SIMD and the build with the least scratch is kept.  (Lane-per-quartet
            // kernels only: 512-register builds of that mode are the tuned form of the large classes; the row-lane
            // mode never goes below two waves per SIMD)
            if (algo == JQC_ALGO_TILE1Q && !(do_j && do_k) && scratch_bytes(code) > 512) {
                long best = scratch_bytes(code);
                for (int minw = (v_minw ? v_minw : 2) - 1; minw >= 1 && best > 0; minw--) {
                    std::vector<std::string> dd = d;
                    bool replaced = false;
                    for (auto& x : dd)
                        if (x.rfind("-DMINW=", 0) == 0) { x = "-DMINW=" + std::to_string(minw); replaced = true; }
                    if (!replaced) dd.push_back("-DMINW=" + std::to_string(minw));
                    std::string c2;
                    rc = build(dd, c2);
                    if (rc) return rc;
                    if ((long)scratch_bytes(c2) < best) { best = scratch_bytes(c2); code.swap(c2); }
                }
            }
            rc = write_code(code, out);
            if (rc) return rc;
        } else {
            int rc = compile_to(src, d, out);
            if (rc) return rc;
        }
    }
    Kernel k;
    k.key = key;
    k.li = li; k.lj = lj; k.lk = lk; k.ll = ll; k.fp32 = fp32; k.algo = algo;
    k.nroots = (li + lj + lk + ll) / 2 + 1;
    if (!compile_only) {
        int rc = load_kernel(out, entry, k);
        if (rc) return rc;
    }
    int h;
    if (it != g_by_key.end()) {
        h = it->second;
        g_kernels[h] = k;
    } else {
        h = (int)g_kernels.size();
        g_kernels.push_back(k);
        g_by_key[key] = h;
    }
    return h;
}

int jqc_jk_launch(int handle, int nao, const void* basis_d, const void* dm_d, double* vj_d, double* vk_d,
                  double omega, const void* quartets_d, const uint32_t* ntasks_d, int64_t ntasks_max, int qstride,
                  int n_dm, void* stream)
{
    KernelView k;
    if (!kernel_view(handle, k)) return fail(-1, "invalid kernel handle %d", handle);
    if (!g_rys64) return fail(-1, "Rys tables not uploaded (jqc_set_rys_tables)");
    if (ntasks_max <= 0) return 0;
    const int n = k.nroots;
    float omega_f = (float)omega;
    const void* cheb = k.fp32 ? (const void*)rys_cheb32(n) : (const void*)rys_cheb64(n);
    const void* large = k.fp32 ? (const void*)rys_large32(n) : (const void*)rys_large64(n);
    void* args[] = {&nao, &basis_d, &dm_d, &vj_d, &vk_d, k.fp32 ? (void*)&omega_f : (void*)&omega,
                    &quartets_d, &ntasks_d, &qstride, &n_dm, &cheb, &large};
    const int block = 256;
    int64_t blocks = (ntasks_max + block - 1) / block;
    const int64_t cap = 256 * 64;  // grid-stride beyond this
    if (blocks > cap) blocks = cap;
    HIP_OK(hipModuleLaunchKernel(k.fn, (unsigned)blocks, 1, 1, block, 1, 1, 0, (hipStream_t)stream, args, nullptr));
    return 0;
}

int jqc_jk_tile_launch(int handle, int nao, const void* basis_d, const void* dm_d, double* vj_d, double* vk_d,
                       double omega, const int32_t* tasks_d, int ntasks, int nblocks, const uint32_t* tpair_sh_d,
                       const float* tpair_q_d, const float* q_cond_d, const float* log_dm_d, int nbas, float cut_lo,
                       float cut_hi, float log_max_dm, int n_dm, uint64_t* counter_d, const int32_t* blk_index_d,
                       const uint32_t* tpair_ao_d, const uint32_t* tpair_pp_d, const void* pair_tab_d, uint64_t* counter32_d,
                       void* stream)
{
    KernelView k;
    if (!kernel_view(handle, k)) return fail(-1, "invalid kernel handle %d", handle);
    if (k.algo != JQC_ALGO_TILE && k.algo != JQC_ALGO_TILE1Q && k.algo != JQC_ALGO_TILE512)
        return fail(-1, "handle %d is not a tile kernel", handle);
    if (!g_rys64) return fail(-1, "Rys tables not uploaded (jqc_set_rys_tables)");
    if (nbas > 46340) return fail(-1, "nbas = %d: the kernels index nbas x nbas tables with 32-bit integers (limit 46340)", nbas);
    if (ntasks <= 0 || nblocks <= 0) return 0;
    // the AQL dispatch packet holds the grid size in WORK-ITEMS as 32 bits: a larger grid is truncated without an error
    if ((unsigned long long)nblocks * 512ull >= (1ull << 32))
        return fail(-1, "%d workgroups exceed the 32-bit work-item count of one launch: use longer ket chunks", nblocks);
    const int n = k.nroots;
    float omega_f = (float)omega;
    const void* cheb = k.fp32 ? (const void*)rys_cheb32(n) : (const void*)rys_cheb64(n);
    const void* large = k.fp32 ? (const void*)rys_large32(n) : (const void*)rys_large64(n);
    void* args[] = {&nao, &basis_d, &dm_d, &vj_d, &vk_d, k.fp32 ? (void*)&omega_f : (void*)&omega, &tasks_d, &ntasks,
                    &tpair_sh_d, &tpair_q_d, &q_cond_d, &log_dm_d, &nbas, &cut_lo, &cut_hi, &log_max_dm, &n_dm,
                    &cheb, &large, &counter_d, &blk_index_d, &tpair_ao_d, &tpair_pp_d, &pair_tab_d, &counter32_d};
    const unsigned threads = k.algo == JQC_ALGO_TILE512 ? 512 : 256;
    HIP_OK(hipModuleLaunchKernel(k.fn, (unsigned)nblocks, 1, 1, threads, 1, 1, 0, (hipStream_t)stream, args, nullptr));
    return 0;
}

int jqc_pair_ntrip(int lk, int ll)
{
    int n = 0;
    for (int t = lk; t <= lk + ll; t++) n += (t + 1) * (t + 2) / 2;
    return n;
}

int jqc_gen_pair_vj_kernel(int li, int lj, int lk, int ll, int rys_lr, int compile_only)
{
    std::lock_guard<std::mutex> lk_(g_mu);
    if (li > JQC_LMAX || lk > JQC_LMAX || lj > li || ll > lk || lj < 0 || ll < 0)
        return fail(-1, "unsupported pair class (%d%d|%d%d): need LMAX >= li >= lj, LMAX >= lk >= ll", li, lj, lk, ll);
    char key[96];
    snprintf(key, sizeof key, "pairvj_%d%d%d%d_lr%d", li, lj, lk, ll, rys_lr);
    auto it = g_by_key.find(key);
    if (it != g_by_key.end()) {
        const Kernel& k = g_kernels[it->second];
        if (k.algo == -JQC_ALGO_PAIRVJ) return fail(-4, "pair_vj %s needs scratch: class stays on the tiled kernels", key);
        if (compile_only || k.fn) return it->second;
    }
    char entry[64];
    snprintf(entry, sizeof entry, "pair_vj_%d%d%d%d", li, lj, lk, ll);
    const std::string out = g_cache_dir + "/" + key + "_" + g_pair_tag + ".hsaco";
    const std::string none = out + ".scratch";            // marker: the build spills, never load it
    Kernel k;
    k.key = key;
    k.li = li; k.lj = lj; k.lk = lk; k.ll = ll; k.algo = JQC_ALGO_PAIRVJ;
    k.nroots = (li + lj + lk + ll) / 2 + 1;
    if (!file_exists(out) && !file_exists(none)) {
        std::vector<std::string> d = {"-DLI=" + std::to_string(li), "-DLJ=" + std::to_string(lj),
                                      "-DLK=" + std::to_string(lk), "-DLL=" + std::to_string(ll),
                                      "-DRYS_LR=" + std::to_string(rys_lr), "-DFP32=0", std::string("-DKNAME=") + entry};
        // register budget: as many waves per SIMD as the class allows without scratch (latency of the scalar ket loads)
        std::string code;
        bool ok = false;
        for (int minw : {4, 2, 1}) {
            d.push_back("-DMINW=" + std::to_string(minw));
            int rc = compile_code("pair_vj.hip", d, code);
            d.pop_back();
            if (rc) return rc;
            if (scratch_bytes(code) == 0) { ok = true; break; }
        }
        int rc = write_code(ok ? code : std::string("scratch\n"), ok ? out : none);
        if (rc) return rc;
    }
    if (!file_exists(out)) {
        k.algo = -JQC_ALGO_PAIRVJ;
        if (it == g_by_key.end()) { g_kernels.push_back(k); g_by_key[key] = (int)g_kernels.size() - 1; }
        return fail(-4, "pair_vj %s needs scratch: class stays on the tiled kernels", key);
    }
    if (!compile_only) {
        int rc = load_kernel(out, entry, k);
        if (rc) return rc;
    }
    int h;
    if (it != g_by_key.end()) {
        h = it->second;
        g_kernels[h] = k;
    } else {
        h = (int)g_kernels.size();
        g_kernels.push_back(k);
        g_by_key[key] = h;
    }
    return h;
}

// quartets one workgroup of the cooperative gradient kernel takes per pass (the constants G / QBYTES of jk_grad.hip)
static int grad_quartets_per_pass(int li, int lj, int lk, int ll)
{
    auto nf = [](int l) { return (l + 1) * (l + 2) / 2; };
    const int t = nf(li) * nf(lj);
    const int gsz = (li + 2) * (lj + 2) * (lk + 2) * (ll + 1);
    const int nkt = (lk + 1) * (ll + 1);
    const int gsb = (li + 1) * (lj + 1) * (nkt % 4 == 0 ? nkt + 1 : nkt);          // (CGSB: record index space with the padded j stride)
    const int nrg = (li + lj + lk + ll + 1) / 2 + 1;
    static const bool a1_map_off = getenv("JQC_EXTRA_DEFS") && strstr(getenv("JQC_EXTRA_DEFS"), "-DGRAD_A1_MAP=0");      // (A/B builds)
    static const char* qpad_s = getenv("JQC_EXTRA_DEFS") ? strstr(getenv("JQC_EXTRA_DEFS"), "-DQPAD=") : nullptr;                 // (A/B builds)
    const int qbytes = (3 * (gsz + 4 * gsb) + (qpad_s ? atoi(qpad_s + 7) : 2) + ((6 * gsz) % 32 == 0 ? 1 : 0) + 2 * nrg + nf(lk) * nf(ll) + 9 + (a1_map_off ? 0 : 24)) * 8;      // (+ QPAD, EPAD: bank padding of sQ / sExt, + NPAR: sPar)
    auto gcap = [&](int budget) { const int g = budget / qbytes; return g < 256 / t ? (g < 1 ? 1 : g) : 256 / t; };
    static const bool two_wg_off = getenv("JQC_EXTRA_DEFS") && strstr(getenv("JQC_EXTRA_DEFS"), "-DGRAD_TWO_WG=0");
    const bool two_wg = !two_wg_off && nf(lk) * nf(ll) <= 18 && 4 * gcap(72 * 1024) >= 3 * gcap(150 * 1024);
    return two_wg ? gcap(72 * 1024) : gcap(150 * 1024);
}

int jqc_gen_jk_grad_kernel(int li, int lj, int lk, int ll, int rys_lr, int compile_only)
{
    std::lock_guard<std::mutex> lk_(g_mu);
    if (li > JQC_LMAX || lj > li || lk > li || ll > lk || li < 0 || lj < 0 || lk < 0 || ll < 0)
        return fail(-1, "unsupported angular class (%d%d|%d%d): need LMAX>=li>=lj, li>=lk>=ll", li, lj, lk, ll);
    // Which form (jk_grad.hip): the cooperative one (GRAD_COOP) where it measured faster on the 112-atom def2-TZVPP gradient
    // (36 of the 65 s..f classes, profiles/r05_grad_forms_per_class_112atoms_tzvpp.txt -- round 5, after the one-quartet-per-lane
    // form got its LDS per-atom table, block-wise density reads and the W form: everything from 324 integrals with a p, d or f
    // ket pair, and (dp|pp)), the one-quartet-per-lane form for the rest.  g classes (not in that workload): by the size of the ket block.
    // JQC_GRAD_COOP=0 / 1 forces one form on every class (A/B).
    static const int coop_env = getenv("JQC_GRAD_COOP") ? atoi(getenv("JQC_GRAD_COOP")) : -1;
    static const char* const kCoopWins[] = {"2022", "2111", "2121", "2122", "2211", "2220", "2221", "2222", "3021", "3022", "3031", "3032",
        "3033", "3111", "3121", "3122", "3130", "3131", "3132", "3133", "3211", "3220", "3221", "3222", "3230", "3231", "3232", "3233",
        "3311", "3320", "3321", "3322", "3330", "3331", "3332", "3333"};
    int coop = coop_env;
    if (coop < 0) {
        if (li <= 3) {
            char c4[8];
            snprintf(c4, sizeof c4, "%d%d%d%d", li, lj, lk, ll);
            coop = 0;
            for (const char* w : kCoopWins) coop |= !strcmp(w, c4);
        } else {
            coop = ((lk + 1) * (lk + 2) / 2) * ((ll + 1) * (ll + 2) / 2) >= 18;
        }
    }
    char key[96];
    snprintf(key, sizeof key, coop ? "jkgrad_%d%d%d%d_lr%d_coop" : "jkgrad_%d%d%d%d_lr%d", li, lj, lk, ll, rys_lr);
    auto it = g_by_key.find(key);
    if (it != g_by_key.end() && (compile_only || g_kernels[it->second].fn)) return it->second;
    char entry[64];
    snprintf(entry, sizeof entry, "jk_grad_%d%d%d%d", li, lj, lk, ll);
    const std::string out = g_cache_dir + "/" + key + "_" + g_grad_tag + ".hsaco";
    if (!file_exists(out)) {
        std::vector<std::string> gd = {"-DLI=" + std::to_string(li), "-DLJ=" + std::to_string(lj),
                                       "-DLK=" + std::to_string(lk), "-DLL=" + std::to_string(ll),
                                       "-DRYS_LR=" + std::to_string(rys_lr), "-DFP32=0", "-DDO_J=1", "-DDO_K=1",
                                       "-DGRAD_COOP=" + std::to_string(coop), std::string("-DKNAME=") + entry};
        // the host's copy of the kernel's quartets-per-pass arithmetic (grad_quartets_per_pass) is pinned at compile time: the
        // cooperative build static_asserts EXPECT_G == G, so a drift between the two stops the build instead of mis-sizing a grid
        if (coop) gd.push_back("-DEXPECT_G=" + std::to_string(grad_quartets_per_pass(li, lj, lk, ll)));
        int rc = compile_to("jk_grad.hip", gd, out);
        if (rc) return rc;
    }
    Kernel k;
    k.key = key;
    k.li = li; k.lj = lj; k.lk = lk; k.ll = ll; k.algo = JQC_ALGO_JKGRAD;
    k.nroots = (li + lj + lk + ll + 1) / 2 + 1;          // one derivative: one more order
    if (coop) k.per_wg = grad_quartets_per_pass(li, lj, lk, ll);
    if (!compile_only) {
        int rc = load_kernel(out, entry, k);
        if (rc) return rc;
    }
    int h;
    if (it != g_by_key.end()) {
        h = it->second;
        g_kernels[h] = k;
    } else {
        h = (int)g_kernels.size();
        g_kernels.push_back(k);
        g_by_key[key] = h;
    }
    return h;
}

int jqc_jk_grad_launch(int handle, int nao, const double* basis_d, const double* dm_d, int n_dm, double* grad_d,
                       const int32_t* shell_atom_d, int natm, int nrep, double j_factor, double k_factor, double omega,
                       const void* quartets_d, const uint32_t* ntasks_d, int64_t ntasks_max, int qstride, void* stream)
{
    KernelView k;
    if (!kernel_view(handle, k)) return fail(-1, "invalid kernel handle %d", handle);
    if (k.algo != JQC_ALGO_JKGRAD) return fail(-1, "handle %d is not a gradient kernel", handle);
    if (!g_rys64) return fail(-1, "Rys tables not uploaded (jqc_set_rys_tables)");
    if (n_dm < 1 || n_dm > 2) return fail(-1, "n_dm = %d: one (closed shell) or two (alpha, beta) densities", n_dm);
    if (k.nroots > (int)g_rys_host[0]) return fail(-1, "class needs %d Rys roots, tables hold %d", k.nroots, (int)g_rys_host[0]);
    if (ntasks_max <= 0) return 0;
    if (nrep < 1) nrep = 1;
    const double* cheb = rys_cheb64(k.nroots);
    const double* large = rys_large64(k.nroots);
    void* args[] = {&nao, &basis_d, &dm_d, &n_dm, &grad_d, &shell_atom_d, &natm, &nrep, &j_factor, &k_factor, &omega,
                    &quartets_d, &ntasks_d, &qstride, &cheb, &large};
    const int block = 256;
    const int per = k.per_wg > 0 ? k.per_wg : block;          // quartets one workgroup takes per pass
    int64_t blocks = (ntasks_max + per - 1) / per;
    const int64_t cap = 256 * 64;  // grid-stride beyond this
    if (blocks > cap) blocks = cap;
    HIP_OK(hipModuleLaunchKernel(k.fn, (unsigned)blocks, 1, 1, block, 1, 1, 0, (hipStream_t)stream, args, nullptr));
    return 0;
}

int jqc_pair_ket_density(const double* basis_d, const double* dm_d, int nao, const uint32_t* ket_pairs_d, int n_ket, int lk,
                         int ll, double* E_d, float* ld_d, void* stream)
{
    if (n_ket <= 0) return 0;
    if (lk > JQC_LMAX || ll > lk || ll < 0) return fail(-1, "unsupported ket class (%d%d)", lk, ll);
    hipLaunchKernelGGL(pair_ket_density_kernel, dim3((n_ket + 63) / 64), dim3(64), 0, (hipStream_t)stream, basis_d, dm_d, nao,
                       ket_pairs_d, n_ket, lk, ll, E_d, ld_d);
    HIP_OK(hipGetLastError());
    return 0;
}

int jqc_pair_vj_launch(int handle, int nao, const double* basis_d, const double* E_d, double* vj_d, double omega,
                       const uint32_t* bra_pairs_d, int n_bra, const float* bra_q_d, const double* bra_tab_d,
                       const uint32_t* ket_pairs_d, const float* ket_q_d, const float* ket_ld_d, const double* ket_tab_d,
                       const int32_t* ket_seg_d, int nseg, float log_cut, float log_max_dm, int npi, int npj, int nsplit,
                       uint64_t* counter_d, void* stream)
{
    KernelView k;
    if (!kernel_view(handle, k)) return fail(-1, "invalid kernel handle %d", handle);
    if (k.algo != JQC_ALGO_PAIRVJ) return fail(-1, "handle %d is not a pair_vj kernel", handle);
    if (!g_rys64) return fail(-1, "Rys tables not uploaded (jqc_set_rys_tables)");
    if (n_bra <= 0 || nseg <= 0) return 0;
    if (nsplit < 1) nsplit = 1;
    const double* cheb = rys_cheb64(k.nroots);
    const double* large = rys_large64(k.nroots);
    void* args[] = {&nao, &basis_d, &E_d, &vj_d, &omega, &bra_pairs_d, &n_bra, &bra_q_d, &bra_tab_d, &ket_pairs_d, &ket_q_d,
                    &ket_ld_d, &ket_tab_d, &ket_seg_d, &nseg, &log_cut, &log_max_dm, &npi, &npj, &cheb, &large, &counter_d};
    HIP_OK(hipModuleLaunchKernel(k.fn, (unsigned)((n_bra + 255) / 256), (unsigned)nsplit, 1, 256, 1, 1, 0, (hipStream_t)stream,
                                 args, nullptr));
    return 0;
}

int jqc_int1e(const double* basis_d, const int32_t* ao_loc_d, const uint32_t* pairs_d, int npairs, const double* atoms_d,
              int natm, int nao, double* S_d, double* T_d, double* V_d, void* stream)
{
    if (npairs <= 0) return 0;
    if (!g_rys64) return fail(-1, "Rys tables not uploaded (jqc_set_rys_tables)");
    // (the device blob starts with its own header: offsets of cheb_n / large_n, joltqc_amd/backend/rys.py)
    hipLaunchKernelGGL(int1e_kernel, dim3((npairs + 63) / 64), dim3(64), 0, (hipStream_t)stream, basis_d, ao_loc_d, pairs_d,
                       npairs, atoms_d, natm, nao, (const double*)g_rys64, (const double*)g_rys64, S_d, T_d, V_d);
    HIP_OK(hipGetLastError());
    return 0;
}

int jqc_ecp_scalar(const double* basis_d, int nao, const int32_t* tasks_d, int ntasks, const double* ecp_xyz_d, const int32_t* ecp_loc_d,
                   const double* ecp_terms_d, const double* rgrid_d, const double* wgrid_d, int nr, const double* ylm_d, double* mat_d,
                   int symmetric, int lmax_shell, void* stream)
{
    if (ntasks <= 0) return 0;
    if (nr <= 0) return fail(-1, "jqc_ecp_scalar: empty radial grid");
    if (lmax_shell > 6) return fail(-1, "jqc_ecp_scalar: shells up to l = 6 (second derivatives of g shells)");
    // two instantiations of the kernel (ecp_kernels.inc): shells up to h with three radial points per chunk, up to i with two
    if (lmax_shell <= 5)
        hipLaunchKernelGGL(ecp_scalar_kernel_l5, dim3(ntasks), dim3(256), 0, (hipStream_t)stream, basis_d, nao, tasks_d, ecp_xyz_d,
                           ecp_loc_d, ecp_terms_d, rgrid_d, wgrid_d, nr, ylm_d, mat_d, symmetric);
    else
        hipLaunchKernelGGL(ecp_scalar_kernel_l6, dim3(ntasks), dim3(256), 0, (hipStream_t)stream, basis_d, nao, tasks_d, ecp_xyz_d,
                           ecp_loc_d, ecp_terms_d, rgrid_d, wgrid_d, nr, ylm_d, mat_d, symmetric);
    HIP_OK(hipGetLastError());
    return 0;
}

int jqc_screen_jk_tasks(const int32_t* tasks_d, int ntasks, int nblocks, const uint32_t* pair_sh_d,
                        const float* pair_q_d, const float* log_dm_d, int nbas, int do_j, int do_k,
                        float log_cutoff_fp32, float log_cutoff_fp64, float log_max_dm, void* queue_d,
                        const int64_t* region_d, uint32_t* counters_d, void* stream)
{
    if (ntasks <= 0 || nblocks <= 0) return 0;
    hipLaunchKernelGGL(screen_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, tasks_d, ntasks, pair_sh_d,
                       pair_q_d, log_dm_d, nbas, do_j, do_k, log_cutoff_fp32, log_cutoff_fp64, log_max_dm,
                       (ushort4*)queue_d, (const long long*)region_d, counters_d);
    HIP_OK(hipGetLastError());
    return 0;
}

int jqc_pair_table(const double* basis_d, const uint32_t* tpair_sh_d, const uint32_t* tpair_wij_d,
                   const uint32_t* pp_off_d, int npairs, double* out_d, void* stream)
{
    if (npairs <= 0) return 0;
    hipLaunchKernelGGL(pair_table_kernel, dim3(npairs), dim3(64), 0, (hipStream_t)stream, basis_d, tpair_sh_d,
                       tpair_wij_d, pp_off_d, out_d);
    HIP_OK(hipGetLastError());
    return 0;
}

int jqc_shell_block_max(const double* mat_d, int n_dm, int nao, const int32_t* ao_loc_d, int nbas, float* out_d,
                        void* stream)
{
    const int n = nbas * nbas;
    hipLaunchKernelGGL(shell_block_max_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, mat_d, n_dm,
                       nao, ao_loc_d, nbas, out_d);
    HIP_OK(hipGetLastError());
    return 0;
}

int jqc_schwarz(int li, int lj, const double* basis_d, const uint32_t* pair_sh_d, int npairs, double omega,
                double* out_d, void* stream)
{
    if (npairs <= 0) return 0;
    if (!g_rys64) return fail(-1, "Rys tables not uploaded (jqc_set_rys_tables)");
    hipFunction_t sfn = nullptr;
    int snroots = 0;
    {
        std::lock_guard<std::mutex> lk_(g_mu);
        const int lr = omega > 0 ? 1 : 0;
        char key[64];
        snprintf(key, sizeof key, "schwarz_%d%d_lr%d", li, lj, lr);
        auto it = g_by_key.find(key);
        if (it == g_by_key.end()) {
            const std::string out = g_cache_dir + "/" + key + "_" + g_src_tag + ".hsaco";
            if (!file_exists(out)) {
                int rc = compile_to("schwarz.hip", {"-DLI=" + std::to_string(li), "-DLJ=" + std::to_string(lj),
                                                    "-DRYS_LR=" + std::to_string(lr)}, out);
                if (rc) return rc;
            }
            Kernel k;
            k.key = key;
            k.nroots = li + lj + 1;
            int rc = load_kernel(out, "schwarz", k);
            if (rc) return rc;
            g_kernels.push_back(k);
            g_by_key[key] = (int)g_kernels.size() - 1;
            it = g_by_key.find(key);
        }
        sfn = g_kernels[it->second].fn;
        snroots = g_kernels[it->second].nroots;
    }
    const int n = snroots;
    const double* cheb = rys_cheb64(n);
    const double* large = rys_large64(n);
    void* args[] = {&basis_d, &pair_sh_d, &npairs, &omega, &out_d, &cheb, &large};
    HIP_OK(hipModuleLaunchKernel(sfn, (unsigned)((npairs + 63) / 64), 1, 1, 64, 1, 1, 0, (hipStream_t)stream, args,
                                 nullptr));
    return 0;
}


// ------------------------------------------------------------------------------------------------ DFT grid path
int jqc_dft_ao_screen(const double* coords_d, int ngrids, const double* basis_d, const int32_t* ao_loc_d, int nbas,
                      float log_cutoff, uint16_t* shell_list_d, int32_t* row_of_d, int32_t* nshl_d, int32_t* nrow_d,
                      float* shell_la_d, void* stream)
{
    if (ngrids % NG) return fail(-1, "ngrids (%d) must be a multiple of %d", ngrids, NG);
    if (nbas > DFT_NBMAX) return fail(-1, "more than %d shells are not supported by the grid screening kernel", DFT_NBMAX);
    if (ngrids == 0) return 0;
    hipLaunchKernelGGL(ao_screen_kernel, dim3(ngrids / NG), dim3(256), 0, (hipStream_t)stream, coords_d, ngrids, basis_d,
                       ao_loc_d, nbas, log_cutoff, shell_list_d, row_of_d, nshl_d, nrow_d, shell_la_d);
    HIP_OK(hipGetLastError());
    return 0;
}

int jqc_dft_eval_ao(const double* coords_d, int ngrids, const double* basis_d, int nbas, int blk0, int nblk,
                    const uint16_t* shell_list_d, const int32_t* row_of_d, const int32_t* nshl_d, const int32_t* nrow_d,
                    const int64_t* row_base_d, int ncomp, int64_t comp_stride, double* ws_d, int32_t* ao_idx_d,
                    const float* shell_la_d, float* row_la_d, void* stream)
{
    if (nblk <= 0) return 0;
    hipLaunchKernelGGL(eval_ao_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, coords_d, ngrids, basis_d, nbas,
                       blk0, shell_list_d, row_of_d, nshl_d, nrow_d, (const long long*)row_base_d, ncomp,
                       (long long)comp_stride, ws_d, ao_idx_d, shell_la_d, row_la_d);
    HIP_OK(hipGetLastError());
    return 0;
}

int jqc_dft_rho(int blk0, int nblk, int ngrids, const int32_t* nrow_d, const int64_t* row_base_d, int64_t comp_stride,
                const double* ws_d, const int32_t* ao_idx_d, const double* dm_d, int nao, int ndim, double* rho_d,
                const float* row_la_d, float thr64, float thr32, const int32_t* order_d, void* stream)
{
    if (nblk <= 0) return 0;
    if (ndim != 1 && ndim != 4 && ndim != 5) return fail(-1, "ndim must be 1 (LDA), 4 (GGA) or 5 (meta-GGA)");
    static const int mgga_one_launch = getenv("JQC_RHO_MGGA") ? atoi(getenv("JQC_RHO_MGGA")) : 0;      // 1: the <1,4> form of rounds 1-2
    if (ndim > 4 && mgga_one_launch)
        hipLaunchKernelGGL((rho_mfma_kernel<1, 4>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, blk0, ngrids, nrow_d,
                           (const long long*)row_base_d, (long long)comp_stride, ws_d, ao_idx_d, dm_d, nao, ndim, rho_d, row_la_d, thr64, thr32, order_d);
    else if (ndim > 4) {
        // rho, grad rho from C = D phi; tau = 1/2 sum_x (D d_x phi) . d_x phi in two more launches (same stream: each adds to rho[4])
        hipLaunchKernelGGL((rho_mfma_kernel<4, 1, 0>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, blk0, ngrids, nrow_d,
                           (const long long*)row_base_d, (long long)comp_stride, ws_d, ao_idx_d, dm_d, nao, ndim, rho_d, row_la_d, thr64, thr32, order_d);
        hipLaunchKernelGGL((rho_mfma_kernel<2, 2, 1>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, blk0, ngrids, nrow_d,
                           (const long long*)row_base_d, (long long)comp_stride, ws_d, ao_idx_d, dm_d, nao, ndim, rho_d, row_la_d, thr64, thr32, order_d);
        hipLaunchKernelGGL((rho_mfma_kernel<4, 1, 3>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, blk0, ngrids, nrow_d,
                           (const long long*)row_base_d, (long long)comp_stride, ws_d, ao_idx_d, dm_d, nao, ndim, rho_d, row_la_d, thr64, thr32, order_d);
    } else
        hipLaunchKernelGGL((rho_mfma_kernel<4, 1>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, blk0, ngrids, nrow_d,
                           (const long long*)row_base_d, (long long)comp_stride, ws_d, ao_idx_d, dm_d, nao, ndim, rho_d, row_la_d, thr64, thr32, order_d);
    HIP_OK(hipGetLastError());
    return 0;
}

int jqc_dft_vxc(int blk0, int nblk, int ngrids, const int32_t* nrow_d, const int64_t* row_base_d, int64_t comp_stride,
                const double* ws_d, const int32_t* ao_idx_d, const double* wv_d, int ndim, int nao, double* vmat_d,
                const float* row_la_d, float thr64, float thr32, const int32_t* order_d, void* stream)
{
    if (nblk <= 0) return 0;
    if (ndim != 1 && ndim != 4 && ndim != 5) return fail(-1, "ndim must be 1 (LDA), 4 (GGA) or 5 (meta-GGA)");
    // (GGA: T = phi X^T has no symmetric pass; LDA and meta-GGA use the instantiation that computes theirs on and below the diagonal)
    if (ndim == 4)
        hipLaunchKernelGGL((vxc_mfma_kernel<false>), dim3(nblk), dim3(VXC_THREADS), 0, (hipStream_t)stream, blk0, ngrids, nrow_d,
                           (const long long*)row_base_d, (long long)comp_stride, ws_d, ao_idx_d, wv_d, ndim, nao, vmat_d, row_la_d, thr64, thr32, order_d);
    else
        hipLaunchKernelGGL((vxc_mfma_kernel<true>), dim3(nblk), dim3(VXC_THREADS), 0, (hipStream_t)stream, blk0, ngrids, nrow_d,
                           (const long long*)row_base_d, (long long)comp_stride, ws_d, ao_idx_d, wv_d, ndim, nao, vmat_d, row_la_d, thr64, thr32, order_d);
    HIP_OK(hipGetLastError());
    return 0;
}

int jqc_dft_xcgrad_ao(const double* coords_d, int ngrids, const double* basis_d, int nbas, int blk0, int nblk,
                      const uint16_t* shell_list_d, const int32_t* row_of_d, const int32_t* nshl_d, const int32_t* nrow_d,
                      const int64_t* row_base_d, const double* wv_d, int ndim, int64_t comp_stride, double* ws_d,
                      int32_t* ao_idx_d, const float* shell_la_d, float* row_la_d, void* stream)
{
    if (nblk <= 0) return 0;
    if (ndim != 1 && ndim != 4 && ndim != 5) return fail(-1, "XC gradient: ndim must be 1 (LDA), 4 (GGA) or 5 (meta-GGA)");
    hipLaunchKernelGGL(xcgrad_ao_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, coords_d, ngrids, basis_d, nbas,
                       blk0, shell_list_d, row_of_d, nshl_d, nrow_d, (const long long*)row_base_d, wv_d, ndim,
                       (long long)comp_stride, ws_d, ao_idx_d, shell_la_d, row_la_d);
    HIP_OK(hipGetLastError());
    return 0;
}

int jqc_dft_xcgrad(int blk0, int nblk, const int32_t* nrow_d, const int64_t* row_base_d, int64_t comp_stride,
                   const double* ws_d, const int32_t* ao_idx_d, const double* dm_d, int nao, double* gao_d,
                   const float* row_la_d, float thr, const int32_t* order_d, int ndim, void* stream)
{
    if (nblk <= 0) return 0;
    if (ndim > 4)
        hipLaunchKernelGGL((xcgrad_mfma_kernel<5, 1>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, blk0, nrow_d,
                           (const long long*)row_base_d, (long long)comp_stride, ws_d, ao_idx_d, dm_d, nao, gao_d, row_la_d, thr, order_d);
    else
        hipLaunchKernelGGL((xcgrad_mfma_kernel<2, 2>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, blk0, nrow_d,
                           (const long long*)row_base_d, (long long)comp_stride, ws_d, ao_idx_d, dm_d, nao, gao_d, row_la_d, thr, order_d);
    HIP_OK(hipGetLastError());
    return 0;
}

int jqc_release_scratch()
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& kv : g_vv10_scratch)
        if (kv.second.first) (void)hipFree(kv.second.first);
    g_vv10_scratch.clear();
    return 0;
}

int jqc_vv10(double* F_d, double* U_d, double* W_d, const double* vvcoords_d, const double* coords_d,
             const double* W0p_d, const double* W0_d, const double* K_d, const double* Kp_d, const double* RpW_d,
             int vvngrids, int ngrids, int fp32, void* stream)
{
    if (ngrids % NG || vvngrids % NG) return fail(-1, "VV10 grids must be padded to a multiple of %d", NG);
    if (ngrids == 0) return 0;
    // JQC_VV10_NOUT = 1: the one-point-per-lane form of rounds 1-2 (A/B); default: NOUT outer points per lane
    // (measured, N = 262 144, FP32 inner loop: 1 point per lane 49.0 ms, 2: 46.7 ms, 4: 42.3 ms; profiles/r03_vv10_outer_points_per_lane_1_2_4.txt)
    static const int nout = getenv("JQC_VV10_NOUT") ? atoi(getenv("JQC_VV10_NOUT")) : 4;
    const int nb = ngrids / NG;
    // JQC_VV10_PK = 0: the scalar-source FP32 inner loop (A/B); default: the packed-FP32 form with 4 outer points per lane
    // (N = 262 144: scalar 42.2 ms; packed 2 / 4 / 8 points per lane, inner loop split automatically: 24.7 / 23.2 / 62 ms;
    //  N = 1 048 576: 541 -> 366 ms; profiles/r03_vv10_packed_fp32.txt)
    static const int pk = getenv("JQC_VV10_PK") ? atoi(getenv("JQC_VV10_PK")) : 4;
    if (fp32 && (pk == 2 || pk == 4 || pk == 8)) {
        // split the inner loop over blockIdx.y until the grid has ~8 workgroups per CU (JQC_VV10_SPLIT overrides)
        const int nwg = (nb + pk - 1) / pk, njb = vvngrids / NG;
        static const int split_env = getenv("JQC_VV10_SPLIT") ? atoi(getenv("JQC_VV10_SPLIT")) : 0;
        int nsplit = split_env > 0 ? split_env : (2048 + nwg - 1) / nwg;
        nsplit = std::max(1, std::min(nsplit, njb));
        const int jchunk = ((njb + nsplit - 1) / nsplit) * NG;
        nsplit = (vvngrids + jchunk - 1) / jchunk;
        // nsplit > 1: every share writes its partial F, U, W to a scratch [share][3][ngrids] (at most ~50 MB: nsplit x ngrids stays
        // near 2048 x pk x 256 points) and vv10_reduce_shares adds the shares in order -- no atomics, bitwise reproducible sums
        double* out_d = F_d;
        if (nsplit > 1) {
            std::lock_guard<std::mutex> lk(g_mu);
            const size_t need = (size_t)nsplit * 3 * ngrids;
            auto& buf = g_vv10_scratch[stream];
            if (buf.second < need) {
                // (a buffer in use by queued kernels of this stream is released behind them: hipFree synchronises)
                if (buf.first) (void)hipFree(buf.first);
                buf = {nullptr, 0};
                HIP_OK(hipMalloc((void**)&buf.first, need * sizeof(double)));
                buf.second = need;
            }
            out_d = buf.first;
        }
#define VV10_PK(N, C) hipLaunchKernelGGL((vv10_kernel_pk<N, C>), dim3(nwg, nsplit), dim3(256), 0, (hipStream_t)stream, out_d, U_d, \
                                        W_d, vvcoords_d, coords_d, W0p_d, W0_d, K_d, Kp_d, RpW_d, vvngrids, ngrids, jchunk)
        const bool check = !(fp32 & 2);             // fp32 = 3: the denominator test cannot fail (include/jqc_hip.h)
        if (pk == 8) { if (check) VV10_PK(8, true); else VV10_PK(8, false); }
        else if (pk == 4) { if (check) VV10_PK(4, true); else VV10_PK(4, false); }
        else { if (check) VV10_PK(2, true); else VV10_PK(2, false); }
#undef VV10_PK
        if (nsplit > 1)
            hipLaunchKernelGGL(vv10_reduce_shares, dim3((ngrids + 255) / 256), dim3(256), 0, (hipStream_t)stream, out_d, nsplit, ngrids,
                               F_d, U_d, W_d);
    }
    else if (fp32 && nout == 2)
        hipLaunchKernelGGL((vv10_kernel_n<float, 2>), dim3((nb + 1) / 2), dim3(256), 0, (hipStream_t)stream, F_d, U_d, W_d,
                           vvcoords_d, coords_d, W0p_d, W0_d, K_d, Kp_d, RpW_d, vvngrids, ngrids);
    else if (fp32 && nout == 4)
        hipLaunchKernelGGL((vv10_kernel_n<float, 4>), dim3((nb + 3) / 4), dim3(256), 0, (hipStream_t)stream, F_d, U_d, W_d,
                           vvcoords_d, coords_d, W0p_d, W0_d, K_d, Kp_d, RpW_d, vvngrids, ngrids);
    else if (!fp32 && nout >= 2)
        hipLaunchKernelGGL((vv10_kernel_n<double, 2>), dim3((nb + 1) / 2), dim3(256), 0, (hipStream_t)stream, F_d, U_d, W_d,
                           vvcoords_d, coords_d, W0p_d, W0_d, K_d, Kp_d, RpW_d, vvngrids, ngrids);
    else if (fp32)
        hipLaunchKernelGGL(vv10_kernel<float>, dim3(ngrids / NG), dim3(256), 0, (hipStream_t)stream, F_d, U_d, W_d,
                           vvcoords_d, coords_d, W0p_d, W0_d, K_d, Kp_d, RpW_d, vvngrids, ngrids);
    else
        hipLaunchKernelGGL(vv10_kernel<double>, dim3(ngrids / NG), dim3(256), 0, (hipStream_t)stream, F_d, U_d, W_d,
                           vvcoords_d, coords_d, W0p_d, W0_d, K_d, Kp_d, RpW_d, vvngrids, ngrids);
    HIP_OK(hipGetLastError());
    return 0;
}

}  // extern "C"
