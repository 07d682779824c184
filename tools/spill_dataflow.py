"""Static check of the SGPR spill slots (v_writelane / v_readlane pairs) of one code object: is every reload dominated by its
store?  For each slot (VGPR, lane) with a single store S: a reload R that can be reached from the kernel entry, or from the header
of a loop that contains S, along a path that does not execute S reads a value of another iteration (or none at all).
usage: python tools/spill_dataflow.py file.hsaco   (diagnostics of the wrong-result builds)"""
import collections, re, subprocess, sys
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def load(path):
    txt = subprocess.run([OBJDUMP, "-d", path], capture_output=True, text=True).stdout
    ins = []
    for l in txt.splitlines():
        m = re.match(r"^\s+(\S.*?)\s+// ([0-9A-F]{12}):", l)
        if m:
            ins.append((int(m.group(2), 16), m.group(1)))
    return ins


def successors(ins):
    addr2i = {a: i for i, (a, _) in enumerate(ins)}
    succ = [[] for _ in ins]
    for i, (a, t) in enumerate(ins):
        op = t.split()[0]
        m = re.match(r"s_(branch|cbranch_\w+) (\d+)", t)
        if m:
            off = int(m.group(2))
            off -= 65536 if off >= 32768 else 0
            j = addr2i[a + 4 + 4 * off]
            succ[i].append(j)
            if m.group(1) != "branch" and i + 1 < len(ins):
                succ[i].append(i + 1)
        elif op == "s_endpgm":
            pass
        elif i + 1 < len(ins):
            succ[i].append(i + 1)
    return succ


def reach(succ, start, blocked):
    seen = {start}
    q = collections.deque([start])
    while q:
        u = q.popleft()
        for v in succ[u]:
            if v not in seen and v != blocked:
                seen.add(v)
                q.append(v)
    return seen


def main(path):
    ins = load(path)
    succ = successors(ins)
    W = collections.defaultdict(list)
    R = collections.defaultdict(list)
    for i, (a, t) in enumerate(ins):
        m = re.match(r"v_writelane_b32 (v\d+), (\S+), (\d+)", t)
        if m:
            W[(m.group(1), int(m.group(3)))].append(i)
        m = re.match(r"v_readlane_b32 (s\d+), (v\d+), (\d+)", t)
        if m:
            R[(m.group(2), int(m.group(3)))].append(i)
    # natural loops: back edges u -> h where h dominates u (dominators by the iterative data-flow algorithm, per instruction
    # would be slow: per basic block)
    leaders = {0}
    for u in range(len(ins)):
        if len(succ[u]) != 1 or succ[u][0] != u + 1:
            leaders.update(succ[u])
            if u + 1 < len(ins):
                leaders.add(u + 1)
    leaders = sorted(leaders)
    blk_of = {}
    for b, l in enumerate(leaders):
        end = leaders[b + 1] if b + 1 < len(leaders) else len(ins)
        for i in range(l, end):
            blk_of[i] = b
    nb = len(leaders)
    bsucc = [set() for _ in range(nb)]
    for b, l in enumerate(leaders):
        end = (leaders[b + 1] if b + 1 < nb else len(ins)) - 1
        for v in succ[end]:
            bsucc[b].add(blk_of[v])
    bpred = [set() for _ in range(nb)]
    for b in range(nb):
        for v in bsucc[b]:
            bpred[v].add(b)
    full = (1 << nb) - 1
    dom = [full] * nb
    dom[0] = 1
    changed = True
    while changed:
        changed = False
        for b in range(1, nb):
            d = full
            for q in bpred[b]:
                d &= dom[q]
            d |= 1 << b
            if d != dom[b]:
                dom[b] = d
                changed = True
    loops = collections.defaultdict(set)          # header block -> body blocks
    for u in range(nb):
        for h in bsucc[u]:
            if dom[u] >> h & 1:
                body_b = {h}
                st = [u]
                while st:
                    x = st.pop()
                    if x in body_b:
                        continue
                    body_b.add(x)
                    st.extend(bpred[x])
                loops[h] |= body_b
    headers = sorted(leaders[h] for h in loops)

    def body(h):
        bl = loops[blk_of[h]]
        return {i for i in range(len(ins)) if blk_of[i] in bl}
    bodies = {h: body(h) for h in headers}
    print(f"{path}: {len(ins)} instructions, {len(W)} spill slots, {len(headers)} loops")
    bad = 0
    for slot in sorted(W):
        stores = W[slot]
        if len(stores) != 1:
            print("  slot", slot, "has", len(stores), "stores at", [hex(ins[s][0]) for s in stores], "-- not analysed")
            continue
        s = stores[0]
        free = reach(succ, 0, s)                        # reachable from the entry without executing the store
        for r in R[slot]:
            if r in free:
                print(f"  slot {slot}: reload at {ins[r][0]:#x} ({ins[r][1]}) reachable from the entry without the store at {ins[s][0]:#x}")
                bad += 1
        for h in headers:
            if s not in bodies[h]:
                continue
            free = reach(succ, h, s)
            for r in R[slot]:
                if r in bodies[h] and r in free and r != h:
                    print(f"  slot {slot}: reload at {ins[r][0]:#x} ({ins[r][1]}) reachable from the loop header {ins[h][0]:#x} "
                          f"without the store at {ins[s][0]:#x} ({ins[s][1]})")
                    bad += 1
    print("  suspicious reloads:", bad)


if __name__ == "__main__":
    for p in sys.argv[1:]:
        main(p)
