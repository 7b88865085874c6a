#!/usr/bin/env python3
"""In-order vmcnt queue walk over the gfx950 assembly of a kernel whose vector-memory instructions are hand-written (conv2d_widep_f16.hip).

gfx9 retires loads, LDS-DMA copies, stores and atomics through ONE counter in issue order; `s_waitcnt vmcnt(N)` returns when at most N entries are
left.  The compiler derives N from the instructions it emitted itself and does not see the ones inside inline assembly, so the persistent wide
kernel counts by hand.  This script replays the counter over the generated code in layout order and reports

  * HAZARD: an instruction reads or writes a VGPR that an outstanding load is still going to write (the wait in front of it is too weak);
  * per kernel, how many loads were checked and the deepest queue seen.

Layout order is not the control flow: a conditional block is walked as if it were always taken, which can only make the simulated queue LONGER than
the real one (a counted wait then retires less than the hardware does) -- the walk may report a false hazard, never hide one on the straight path.
Behind an unconditional `s_branch` the code is only reached by jumps whose queue the walk does not know: the queue restarts empty there (counted in the
report as `restarts`; the K-step code of the persistent kernel is straight-line between the tile loop's head and its back-edge, the restarts are the
compiler-generated loops of the statistics fold and the kernel's tail).
Used by tests/test_isa.py; run by hand:  python tools/audit_vmcnt.py file.s [kernel-name-substring]
"""
import re
import sys

VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def is_vmem(op):
    return op.startswith(("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "scratch_", "flat_"))


def audit(asm, only=""):
    """-> list of (kernel, loads, max_depth, hazards[], restarts)"""
    results = []
    kernel, queue, loads, depth, hazards, restarts = None, [], 0, 0, [], 0
    for ln, raw in enumerate(asm.split("\n"), 1):
        line = raw.split(";")[0].strip()
        if not line:
            continue
        m = re.match(r"^(_Z\w+):$", line)
        if m:
            kernel, queue, loads, depth, hazards, restarts = m.group(1), [], 0, 0, [], 0
            continue
        if line.startswith(".Lfunc_end") and kernel:
            if only in kernel:
                results.append((kernel, loads, depth, hazards, restarts))
            kernel = None
            continue
        if kernel is None or line.startswith(".") or line.endswith(":"):
            continue
        op, _, rest = line.partition(" ")
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            queue, restarts = [], restarts + 1
            continue
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", rest)
            if m:
                keep = int(m.group(1))
                if keep < len(queue):
                    queue = queue[len(queue) - keep:] if keep else []
            continue
        touched = vregs(rest)
        for dest, where in queue:
            if dest and dest & touched:
                hazards.append("line %d `%s` touches v%s of the load at line %d still in flight (%d entries queued)" % (ln, line, sorted(dest & touched), where, len(queue)))
        if is_vmem(op):
            dest = set()
            if "load" in op and not rest.rstrip().endswith(" lds") and "_lds_" not in op:
                dest = vregs(rest.split(",")[0])
                loads += 1
            elif "atomic" in op and ("sc0" in rest or "glc" in rest):
                dest = vregs(rest.split(",")[0])
            queue.append((dest, ln))
            depth = max(depth, len(queue))
    return results


def main():
    asm = open(sys.argv[1]).read()
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    bad = 0
    for kernel, loads, depth, hazards, restarts in audit(asm, only):
        print("%s: %d register loads checked, deepest queue %d, %d restart(s), %d hazard(s)" % (kernel[:110], loads, depth, restarts, len(hazards)))
        for h in hazards[:10]:
            print("   ", h)
        bad += len(hazards)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
