#!/usr/bin/env python
"""ISA check for kernels that keep inline-asm loads in flight across statements (conv1d_bsplit.hip, conv1d_bsplit2.hip,
conv1d_gemm_split.hip, conv1d_wgrad_split.hip): hipcc does not know that the destination register of an inline-asm
`global_load_*` is not valid until the matching `s_waitcnt vmcnt`, so any instruction it places in between that READS or
WRITES such a register (a copy at a control-flow join, a spill, re-use for another value) silently uses stale data -- a bug
that only shows when the memory system is loaded.  This script compiles a source to gfx950 assembly and runs a forward
dataflow analysis over every kernel's control-flow graph: each load destination carries the number of younger VMEM
operations, `s_waitcnt vmcnt(n)` retires everything with at least n younger ones (loads return in order), states merge at joins
pessimistically, and any instruction touching a register that may still be in flight is reported.  It also reports, per kernel, how many distinct destination registers
the asm loads use and how many load sites share each (one physical register per logical one is the healthy pattern).

Kernels that land their in-flight loads in NAMED physical registers (conv1d_bsplit.hip / conv1d_bsplit2.hip: `global_load_dword
v208` ... written and read only inside inline asm) are checked exactly instead: no instruction outside an inline-asm block may
touch a register of the reserved range (`reserved_violations`).  That check has no false positives and is the one
tests/test_isa_inflight.py enforces; the dataflow findings for compiler-allocated asm loads are path-insensitive (they
include infeasible paths through the pipelined loops' guards) and are printed for review only.

usage: check_inflight_regs.py <file.hip> [...]      exit status 1 if a reserved-register violation is found"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), r) for r in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def compile_to_asm(src):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
           "-I" + os.path.join(REPO, "include"), "-o", out, src] + os.environ.get("FAC_EXTRA_FLAGS", "").split()
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


LOADS = ("global_load", "buffer_load", "flat_load", "scratch_load")
STORES = ("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic", "flat_atomic")


def transfer(state, ins, report):
    """state: {register: number of younger VMEM operations}; returns the new state (report: list collecting violations)."""
    op = ins.split()[0]
    if op == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", ins)
        if m:
            k = int(m.group(1))
            state = {r: d for r, d in state.items() if d < k}
        return state
    is_load, is_store = op.startswith(LOADS), op.startswith(STORES)
    if is_load and "_lds_" not in op:
        dest = regs_of(ins.split(",")[0])
        touched = regs_of(ins.split(",", 1)[1]) if "," in ins else set()
    else:
        dest, touched = set(), regs_of(ins)
    hit = (touched | dest) & set(state)
    if hit and report is not None:
        report.append((ins, sorted(hit)))
    if is_load or is_store:
        state = {r: d + 1 for r, d in state.items() if d + 1 < 64}
        for r in dest:
            state[r] = 0
    return state


def analyse(body):
    """body: instruction / label lines of one kernel.  Forward dataflow over the control-flow graph (merge = union with the
    youngest age), then one reporting pass."""
    blocks, labels, cur = [], {}, []
    for ln in body:
        if ln.endswith(":") or re.match(r"^\.LBB\w+:", ln):
            if cur:
                blocks.append(cur)
                cur = []
            labels[ln.split(":")[0]] = len(blocks)
            continue
        cur.append(ln)
        if ln.split()[0].startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    succ = []
    for i, b in enumerate(blocks):
        last = b[-1].split() if b else ["nop"]
        nxt = []
        if last[0].startswith(("s_branch", "s_cbranch")) and last[1] in labels:
            nxt.append(labels[last[1]])
        if not last[0].startswith(("s_branch", "s_endpgm", "s_setpc")) and i + 1 < len(blocks):
            nxt.append(i + 1)
        succ.append(nxt)
    instate = [None] * len(blocks)
    instate[0] = {}
    work = [0]
    while work:
        i = work.pop()
        st = dict(instate[i])
        for ins in blocks[i]:
            st = transfer(st, ins, None)
        for j in succ[i]:
            if j >= len(blocks):
                continue
            if instate[j] is None:
                instate[j] = dict(st)
                work.append(j)
            else:
                merged, changed = dict(instate[j]), False
                for r, d in st.items():
                    if r not in merged or d < merged[r]:
                        merged[r] = d
                        changed = True
                if changed:
                    instate[j] = merged
                    work.append(j)
    bad = []
    for i, b in enumerate(blocks):
        if instate[i] is None:
            continue
        st = dict(instate[i])
        for ins in b:
            st = transfer(st, ins, bad)
    return bad


def check(asm_path):
    kernels, cur = {}, None
    for line in open(asm_path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
        elif cur is not None:
            t = line.strip()
            if re.match(r"^\.LBB\w+:", t):
                kernels[cur].append(t.split()[0])
            elif line.startswith("\t") and t and not t.startswith((";", ".")):
                kernels[cur].append(t)
            elif t.startswith(".Lfunc_end"):
                cur = None
    report, bad = {}, []
    for name, body in kernels.items():
        sites, n_asm = {}, 0
        for ins in body:
            if re.match(r"global_load_dword(x\d)? [va]\S+, v\d+, s\[", ins):     # the inline-asm form: vdst, voffset, sbase
                n_asm += 1
                for r in regs_of(ins.split(",")[0]):
                    sites[r] = sites.get(r, 0) + 1
        if n_asm:
            hist = {}
            for n in sites.values():
                hist[n] = hist.get(n, 0) + 1
            report[name] = dict(asm_loads=n_asm, distinct_dest_regs=len(sites), sites_per_reg=hist)
            bad += [(name, ins, hit) for ins, hit in analyse(body)]
    return report, bad


def reserved_violations(asm_path):
    """{kernel: (lowest named landing register, [instructions outside inline asm that touch the reserved range])} for kernels whose
    inline asm names physical VGPRs as load destinations."""
    out, cur, in_asm, body = {}, None, False, []

    def flush():
        if cur is None:
            return
        named = [r for ins, a in body if a and re.match(r"global_load_dword(x\d)? v\d+, v\d+, s\[", ins) for (_, r) in regs_of(ins.split(",")[0])]
        if not named:
            return
        # named landing registers are those written by asm loads AND never defined outside asm; take the contiguous top range
        lo = min(r for r in named if r >= 128) if any(r >= 128 for r in named) else None
        if lo is None:
            return
        bad = [ins for ins, a in body if not a and any(k == "v" and r >= lo for (k, r) in regs_of(ins))]
        out[cur] = (lo, bad)

    for line in open(asm_path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            flush()
            cur, in_asm, body = m.group(1), False, []
            continue
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        elif t.startswith(".Lfunc_end"):
            flush()
            cur = None
        elif cur is not None and line.startswith("\t") and t and not t.startswith((";", ".")):
            body.append((t, in_asm))
    flush()
    return out


def named_lifetime_violations(asm_path):
    """{kernel: [(instruction, registers)]}: the exact form of `reserved_violations`.  A named landing register is LIVE from the
    inline-asm load that writes it to the inline-asm `v_cndmask_b32_e64 dst, 0, vR, mask` that takes the value out; a forward
    may-analysis over the control-flow graph (union at joins) carries the live set, and any other instruction that touches a
    live register -- compiler-generated or another asm load -- is a violation.  Code that only waves without loads in flight
    execute (the MFMA waves' instantiation of the tile walk in conv1d_bsplit.hip, the launch prologue) may use the same
    physical registers freely: it is not reachable from a load site."""
    out = {}
    kernels, cur, in_asm = {}, None, False
    for line in open(asm_path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, in_asm = m.group(1), False
            kernels[cur] = []
            continue
        t = line.strip()
        if cur is None:
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        elif t.startswith(".Lfunc_end"):
            cur = None
        elif re.match(r"^\.LBB\w+:", t):
            kernels[cur].append((t.split(":")[0] + ":", False))
        elif line.startswith("\t") and t and not t.startswith((";", ".")):
            kernels[cur].append((t, in_asm))
    for name, body in kernels.items():
        named = {r for ins, a in body if a and re.match(r"global_load_dword(x\d)? v\d+, v\d+, s\[", ins) for (_, r) in regs_of(ins.split(",")[0])}
        named = {r for r in named if r >= 128}
        if not named:
            continue
        blocks, labels, cb = [], {}, []
        for ins, a in body:
            if ins.endswith(":"):
                if cb:
                    blocks.append(cb)
                    cb = []
                labels[ins[:-1]] = len(blocks)
                continue
            cb.append((ins, a))
            if ins.split()[0].startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
                blocks.append(cb)
                cb = []
        if cb:
            blocks.append(cb)
        succ = []
        for i, b in enumerate(blocks):
            last = b[-1][0].split() if b else ["nop"]
            nxt = []
            if last[0].startswith(("s_branch", "s_cbranch")) and last[1] in labels:
                nxt.append(labels[last[1]])
            if not last[0].startswith(("s_branch", "s_endpgm", "s_setpc")) and i + 1 < len(blocks):
                nxt.append(i + 1)
            succ.append([j for j in nxt if j < len(blocks)])

        def step(live, ins, a, report):
            op = ins.split()[0]
            regs = {r for k, r in regs_of(ins) if k == "v" and r in named}
            if a and op.startswith("global_load_dword"):      # (a re-issued load on a path whose take was guarded out is not reported:
                dest = {r for k, r in regs_of(ins.split(",")[0]) if k == "v" and r in named}      # the guards are correlated)
                return live | dest
            if a and op.startswith("v_cndmask_b32"):
                src = {r for k, r in regs_of(ins.split(",", 1)[1]) if k == "v" and r in named}
                return live - src
            if report is not None and (regs & live):
                report.append((ins, sorted(regs & live)))
            return live

        instate = [None] * len(blocks)
        instate[0] = frozenset()
        work = [0]
        while work:
            i = work.pop()
            live = set(instate[i])
            for ins, a in blocks[i]:
                live = step(live, ins, a, None)
            for j in succ[i]:
                if instate[j] is None:
                    instate[j] = frozenset(live)
                    work.append(j)
                elif not live <= instate[j]:
                    instate[j] = frozenset(instate[j] | live)
                    work.append(j)
        bad = []
        for i, b in enumerate(blocks):
            if instate[i] is None:
                continue
            live = set(instate[i])
            for ins, a in b:
                live = step(live, ins, a, bad)
        out[name] = bad
    return out


def spill_counts(asm_path):
    """{kernel: vgpr_spill_count} from the kernel metadata of the assembly."""
    out, name = {}, None
    for line in open(asm_path):
        m = re.match(r"\s*\.name:\s+(\S+)", line)
        if m:
            name = m.group(1)
        m = re.match(r"\s*\.vgpr_spill_count:\s+(\d+)", line)
        if m and name:
            out[name] = int(m.group(1))
    return out


def main():
    rc = 0
    for src in sys.argv[1:]:
        asm = compile_to_asm(src)
        rep, bad = check(asm)
        res = reserved_violations(asm)
        print(os.path.basename(src))
        spills = {k: v for k, v in spill_counts(asm).items() if v > 32}
        for k, v in spills.items():
            print(f"   REGISTER SPILLS: {k[:70]} spills {v} VGPRs to scratch")
            rc = 1
        for k, (lo, viol) in res.items():
            print(f"   {k[:70]}: landing registers v{lo}..v255 named in inline asm, {len(viol)} other instruction(s) touch them")
            for ins in viol[:10]:
                print("      RESERVED-REGISTER VIOLATION |", ins)
            if viol:
                rc = 1
        for k, v in rep.items():
            if k not in res:
                print("  ", k[:70], v)
        shown = [b for b in bad if b[0] not in res]
        for name, ins, hit in shown[:12]:
            print("   possible (path-insensitive)", name[:50], "|", ins, "| in flight:", hit)
        if len(shown) > 12:
            print(f"   ... {len(shown) - 12} more")
    sys.exit(rc)


if __name__ == "__main__":
    main()
