"""Development helper: inserts the QUAD compute path (one quartet per DPP quad of lanes) into a copy of jk_tile.hip.
usage: python tools/dev/apply_quad_patch.py <path to jk_tile.hip>   (idempotent: refuses a file that already has it)"""
import sys
p = sys.argv[1]
s = open(p).read()
assert "#ifndef QUAD" not in s, "already patched"

s = s.replace('''#ifndef ORED
#define ORED 0''', '''#ifndef QUAD
#define QUAD 0      // lane-per-quartet mode for classes with a p shell (index X = the last p among l, k, j, i): a quartet is worked on by
                    // ONE QUAD of lanes (DPP quad = 4 consecutive lanes), 64 quartets per pass of a workgroup.  Lanes c = 0, 1, 2 of
                    // the quad own Cartesian axis c: lane c runs the transfer + horizontal recurrences of axis c only (a third of
                    // the recurrence work, nothing redundant) and owns the integrals whose X component is p_c, i.e. a third of the
                    // integral block.  Such an integral is  g_c[X-power 1] * g_(c+1)[X-power 0] * g_(c+2)[X-power 0]:  the lane needs its
                    // own 1-D array and the X-power-0 HALF of its two neighbours' arrays, which it fetches register to register with
                    // two quad-permute DPP moves per double (no LDS array, no barrier).  Every lane enumerates its block in axes
                    // ROTATED so that its own axis comes first (then the code is the same for the three lanes); the rotated
                    // component labels are turned back into AO offsets by compile-time permutation tables selected by c.
                    // Each lane evaluates ONE Rys root per primitive combination (lane r of the quad: root r, so the fourth lane
                    // is of use when there are four roots) and the root in work is broadcast inside the quad.  Purpose: the
                    // 100-180-integral classes fit 256 registers (no AGPR copies, second workgroup per CU); price: the fourth
                    // lane idles otherwise, and blocks that do not carry X are added to the LDS tiles by three lanes.
#endif
#ifndef ORED
#define ORED 0''', 1)

helpers = r'''
#if QUAD
// split index X and the rotated-component -> AO-component tables
constexpr int XS = LL == 1 ? 3 : LK == 1 ? 2 : LJ == 1 ? 1 : LI == 1 ? 0 : -1;
static_assert(XS >= 0, "QUAD: the class needs a p shell");
static_assert(TILE_1Q && !MIXED && NROOTS <= 4, "QUAD: lane-per-quartet builds up to four Rys roots");
constexpr int NXI = XS == 0 ? 1 : NFI, NXJ = XS == 1 ? 1 : NFJ, NXK = XS == 2 ? 1 : NFK, NXL = XS == 3 ? 1 : NFL;
constexpr int NINTQ = NXI * NXJ * NXK * NXL;              // integrals per lane
// chunks of the lane's block over the components of a second index (QY = 0..3 = i, j, k, l; -1: none): QNCH passes, each with its own
// evaluation of the recurrences and roots, for the classes whose third of the block would not fit the registers
#ifndef QNCH
#define QNCH 1
#endif
#ifndef QY
#define QY (-1)
#endif

static_assert(QNCH == 1 || (QY >= 0 && QY <= 3 && QY != XS), "QUAD chunks: over an index other than the split one");
constexpr int CI = QY == 0 ? NXI / QNCH : NXI, CJ = QY == 1 ? NXJ / QNCH : NXJ, CK = QY == 2 ? NXK / QNCH : NXK, CL = QY == 3 ? NXL / QNCH : NXL;
static_assert(CI * CJ * CK * CL * QNCH == NINTQ, "QUAD chunks must divide the component count of their index");
constexpr int NINTC = CI * CJ * CK * CL;                  // integrals per lane and chunk
constexpr int GS_X = XS == 0 ? GS_I : XS == 1 ? GS_J : XS == 2 ? GS_K : GS_L;
// component n of a shell of angular momentum l, read with the axes rotated by c (own axis first): exponents (p, q, r) on
// (a0, a1, a2) = (c, c+1, c+2) mod 3  ->  index of the Cartesian component with those exponents on (x, y, z)
constexpr int rot_comp(const int l, const int c, const int n)
{
    const CartPow p = cart_pow(l, n);
    const int ex = c == 0 ? p.x : c == 1 ? p.z : p.y;
    const int ey = c == 0 ? p.y : c == 1 ? p.x : p.z;
    return (l - ex) * (l - ex + 1) / 2 + (l - ex - ey);
}
// quad permutes (DPP): lane c of a quad reads lane (c + 1) % 3 / (c + 2) % 3 (lane 3 mirrors lane 0)
#define DPP_ROT1 0x49   /* quad_perm:[1,2,0,1] */
#define DPP_ROT2 0x92   /* quad_perm:[2,0,1,2] */
template <int CTRL> __device__ __forceinline__ double dpp_quad(const double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ float dpp_quad(const float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
// value of lane r of the quad (r wave-uniform)
__device__ __forceinline__ real quad_bcast(const real v, const int r)
{
    switch (r) {
    case 0: return dpp_quad<0x00>(v);
    case 1: return dpp_quad<0x55>(v);
    case 2: return dpp_quad<0xaa>(v);
    default: return dpp_quad<0xff>(v);
    }
}
#endif

// Staging is split into "issue every global load"'''
assert s.count('\n// Staging is split into "issue every global load"') == 1
s = s.replace('\n// Staging is split into "issue every global load"', helpers, 1)

quad_body = open(__file__.replace("apply_quad_patch.py", "quad_body.inc")).read()
marker = '''#if QIL
            // strided read: lane l takes entry'''
assert s.count(marker) == 1
s = s.replace(marker, quad_body + marker)
endm = '''                STAMP(12);          // (diagnostic) contraction + LDS atomics of this batch
}
'''
assert s.count(endm) == 1
s = s.replace(endm, endm + '#endif  // QUAD\n')
open(p, 'w').write(s)
print("patched", p)
