"""Static check of the synchronisation of the gfx950 code objects (no GPU needed): every `s_barrier` of a kernel must be
reached, on EVERY control-flow path, with the wave's LDS accesses (writes, atomics, reads) retired -- i.e. behind an
`s_waitcnt` whose lgkmcnt is 0 with no `ds_*` memory instruction in between.  `__syncthreads()` normally compiles to exactly that pair; round 5
found one barrier of `frame_bb_kernel` emitted WITHOUT its wait (a wave read claim words while another wave's `ds_or` was still
queued: one wrong frame in 1e5, two rounds under a green suite).  Second rule, for `global_load_lds` (HBM -> LDS without
registers, counted by vmcnt): no path from such a load may reach the same instruction again, or the end of the kernel, without
passing `s_waitcnt vmcnt(0)` and then an `s_barrier` -- the pair that publishes the loaded frame to the other waves.

The analysis is a forward may-dataflow over the control-flow graph rebuilt from `llvm-objdump -d` (instruction addresses and
branch targets are in the listing's comments).  It is conservative: a fact that holds on one path only does not count.
Limits, stated: it proves the WAIT, not that the right data is behind it (addresses are not tracked); a call (only the
self-check builds have any: printf) counts as an LDS access of unknown kind."""
import os
import re
import shutil
import subprocess

LLVM = "/opt/rocm/lib/llvm/bin"

_ADDR = re.compile(r"//\s*([0-9A-Fa-f]{8,16}):")
_TARGET = re.compile(r"<([^>+]+)(?:\+0x([0-9a-fA-F]+))?>\s*$")
# DS instructions that do not touch LDS memory (cross-lane moves through the LDS crossbar).  Everything else -- writes,
# atomics AND reads (a read still queued when another wave's write passes the barrier is the same hazard mirrored) -- must be
# retired at a barrier.
_DS_NOT_A_WRITE = ("ds_bpermute", "ds_permute", "ds_swizzle", "ds_nop", "ds_gws")


def disassemble(lib_path, workdir):
    """{mangled kernel name: [(addr, mnemonic, operands, branch_target_addr or None)]} for every function in every gfx950
    code object bundled in the shared library."""
    shutil.copy(lib_path, os.path.join(workdir, "lib.so"))
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=workdir, check=True, capture_output=True)
    funcs = {}
    for f in sorted(os.listdir(workdir)):
        if "amdgcn" not in f or f.endswith(".s"):
            continue
        txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f], cwd=workdir, check=True,
                             capture_output=True, text=True).stdout
        cur, base = None, {}
        for line in txt.split("\n"):
            m = re.match(r"^([0-9a-f]{16}) <(\S+)>:", line)
            if m:
                cur = m.group(2)
                base[cur] = int(m.group(1), 16)
                funcs[cur] = []
                continue
            if cur is None or not line.startswith("\t"):
                continue
            a = _ADDR.search(line)
            if not a:
                continue
            code = line.split("//")[0].strip()
            parts = code.split(None, 1)
            mn, ops = parts[0], (parts[1] if len(parts) > 1 else "")
            tgt = None
            if mn.startswith("s_cbranch") or mn == "s_branch":
                t = _TARGET.search(line)
                assert t and t.group(1) == cur, ("branch out of its function", cur, line)
                tgt = base[cur] + (int(t.group(2), 16) if t.group(2) else 0)
            funcs[cur].append((int(a.group(1), 16), mn, ops, tgt))
    return funcs


def _waits(mn, ops):
    """(lgkmcnt reaches 0, vmcnt reaches 0) for an s_waitcnt."""
    if mn != "s_waitcnt":
        return False, False
    if "cnt(" in ops:
        return "lgkmcnt(0)" in ops, "vmcnt(0)" in ops
    v = int(ops, 0)                                     # bare immediate (gfx9 encoding)
    return ((v >> 8) & 0xF) == 0, ((v & 0xF) | (((v >> 14) & 3) << 4)) == 0


def check_kernel(insts):
    """-> dict(barriers, violations=[(addr, why)], n_global_load_lds).  State per program point (may-analysis, OR at joins):
    bit 0: an LDS access was issued since the last lgkmcnt(0) wait
    bit 1: a global_load_lds is in flight (no vmcnt(0) wait yet)
    bit 2: ... waited for, but no barrier behind the wait yet (not published to the other waves)"""
    idx = {a: i for i, (a, _, _, _) in enumerate(insts)}
    n = len(insts)
    succ = [[] for _ in range(n)]
    for i, (a, mn, ops, tgt) in enumerate(insts):
        # calls (the self-check builds' printf): the callee returns to the next instruction; whatever it did to LDS counts as
        # an access not yet waited for (transfer()).  s_setpc_b64 = a helper function's return.
        if mn in ("s_endpgm", "s_setpc_b64"):
            continue
        if mn == "s_branch":
            succ[i].append(idx[tgt])
            continue
        if mn.startswith("s_cbranch"):
            succ[i].append(idx[tgt])
        if i + 1 < n:
            succ[i].append(i + 1)
    state_in = [0] * n
    seen = [False] * n
    seen[0] = True
    work = [0]
    gll = [i for i, x in enumerate(insts) if x[1].startswith("global_load_lds")]

    def transfer(i, s):
        _, mn, ops, _ = insts[i]
        if (mn.startswith("ds_") and not mn.startswith(_DS_NOT_A_WRITE)) or mn in ("s_swappc_b64", "s_call_b64"):
            s |= 1
        elif mn.startswith("global_load_lds"):
            s = (s | 2) & ~4
        elif mn == "s_waitcnt":
            lg, vm = _waits(mn, ops)
            if lg:
                s &= ~1
            if vm and (s & 2):
                s = (s & ~2) | 4
        elif mn == "s_barrier":
            s &= ~4
        return s

    while work:
        i = work.pop()
        out = transfer(i, state_in[i])
        for j in succ[i]:
            new = state_in[j] | out
            if new != state_in[j] or not seen[j]:
                state_in[j] = new
                seen[j] = True
                work.append(j)
    violations = []
    barriers = 0
    for i, (a, mn, ops, _) in enumerate(insts):
        if not seen[i]:
            continue
        if mn == "s_barrier":
            barriers += 1
            if state_in[i] & 1:
                violations.append((a, "s_barrier reachable with an LDS access not behind s_waitcnt lgkmcnt(0)"))
    # Rule 2.  From each global_load_lds, with EXACT states per instruction (no merging: a wait on one path must not vouch
    # for another): 2 = in flight, 4 = waited for but not yet behind a barrier, 0 = published; | 8 = an s_barrier was passed
    # while 2 or 4 held (the path left the prefetch's own loop and went round the frame loop).  A violation is reaching the
    # same load again, or the end of the kernel, in a state with 8 set and 2 or 4 still held.
    for g in gll:
        start = 2
        states = {g: {start}}
        stack = [(g, start)]
        bad = None
        while stack and bad is None:
            i, s_in = stack.pop()
            for j in succ[i]:
                mn, ops = insts[j][1], insts[j][2]
                if (j == g or mn == "s_endpgm") and (s_in & 6) and (s_in & 8):
                    bad = "the end of the kernel" if mn == "s_endpgm" else "itself again (the next frame)"
                    break
                if j == g or not (s_in & 6):
                    continue                       # published (or back at the load through its own loop): path closed
                s_out = s_in
                if mn == "s_waitcnt" and _waits(mn, ops)[1] and (s_in & 2):
                    s_out = (s_in & ~2) | 4
                elif mn == "s_barrier":
                    s_out = 0 if (s_in & 4) else (s_in | 8)
                if s_out not in states.setdefault(j, set()):
                    states[j].add(s_out)
                    stack.append((j, s_out))
        if bad:
            violations.append((insts[g][0], "global_load_lds reaches %s past a barrier without s_waitcnt vmcnt(0) followed by s_barrier" % bad))
    return {"barriers": barriers, "violations": violations, "n_global_load_lds": len(gll)}
