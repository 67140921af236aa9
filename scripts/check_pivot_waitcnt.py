#!/usr/bin/env python3
"""Build-time check of an invariant the pivot pipeline of the persistent sweep rests on (kernels.hip, cdp_dma_issue):

The fetch of the next diagonal block's inputs is issued by the T waves DURING the pivot steps of the current block as an
LDS-DMA transfer written out in inline asm (`s_mov_b32 m0` + `global_load_lds_dwordx4 ... sc1`), precisely so that the
compiler's SIInsertWaitcnts pass does not know it is a transfer into LDS.  When it does know (the builtin form,
-DCDP_DMA_BUILTIN) it puts `s_waitcnt vmcnt(0)` in front of every later LDS read that may alias the transfer's target --
the flag and operand reads of the T wave's next pivot steps: a wait for global memory inside the pivot loop, 1.2 us per
block (NOTEBOOK.md, round 4).  Nothing in the source says so; a compiler upgrade or a refactoring that lets the backend
see through the asm would bring the waits back silently (results stay right, the frame gets slower).

The check reads the device assembly (`hipcc -S --cuda-device-only` of kernels.hip) and, in every instantiation of
sweep_persistent_kernel, walks forward from each hand-written DMA block to the next `s_barrier`: no compiler-emitted
(i.e. outside `;;#ASMSTART` .. `;;#ASMEND`) `s_waitcnt` naming `vmcnt(0)` may stand within three instructions in front of
an LDS read there.  (tests/test_build_invariants.py feeds the walker a synthetic listing with such a wait to show that it
is seen, and runs it on the assembly of the product build.)

    python scripts/check_pivot_waitcnt.py kernels.s
"""
import re
import sys


def instructions(lines, start, stop):
    """(line number, text, in_asm) of the instruction lines in [start, stop)"""
    in_asm = False
    for i in range(start, stop):
        t = lines[i].strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        yield i, t, in_asm


def check(path):
    return check_lines(open(path).read().split("\n"))


def check_lines(lines):
    # function bodies of the persistent sweep's instantiations
    bodies = []
    start = None
    for i, l in enumerate(lines):
        if re.match(r"_ZN5rslam23sweep_persistent_kernel\w*:", l):
            start = i
        elif start is not None and l.startswith(".Lfunc_end"):
            bodies.append((start, i))
            start = None
    report = {"functions": len(bodies), "dma_blocks": 0, "violations": []}
    for (b0, b1) in bodies:
        ins = list(instructions(lines, b0, b1))
        n = len(ins)
        k = 0
        while k < n:
            i, t, in_asm = ins[k]
            if in_asm and t.startswith("global_load_lds_dwordx4") and k >= 2 and ins[k - 2][1].startswith("s_mov_b32 m0"):
                report["dma_blocks"] += 1
                # forward to the next barrier
                j = k + 1
                while j < n and not ins[j][1].startswith("s_barrier"):
                    ii, tt, aa = ins[j]
                    if not aa and tt.startswith("s_waitcnt") and re.search(r"vmcnt\(0\)", tt):
                        nxt = [ins[q][1] for q in range(j + 1, min(n, j + 4))]
                        if any(x.startswith("ds_read") or x.startswith("ds_load") for x in nxt):
                            report["violations"].append((ii + 1, tt, nxt))
                    # a following hand-written DMA block restarts the same walk: stop here, the outer loop continues there
                    if aa and tt.startswith("global_load_lds_dwordx4"):
                        break
                    j += 1
            k += 1
    return report


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    rep = check(sys.argv[1])
    uniq = sorted(set(v[0] for v in rep["violations"]))
    print("sweep_persistent_kernel instantiations: %d, hand-written LDS-DMA blocks: %d, compiler waits for vmcnt(0) in front of "
          "LDS reads between a block and the next barrier: %d" % (rep["functions"], rep["dma_blocks"], len(uniq)))
    for v in rep["violations"][:8]:
        print("  line %d: %s   -> %s" % (v[0], v[1], " | ".join(v[2])))
    if rep["functions"] == 0 or rep["dma_blocks"] == 0:
        raise SystemExit("check_pivot_waitcnt: the assembly holds no persistent sweep / no hand-written DMA block: nothing was checked")
    if uniq:
        raise SystemExit("check_pivot_waitcnt: the compiler waits for global memory inside the pivot steps again")


if __name__ == "__main__":
    main()
