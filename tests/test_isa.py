"""Build-time checks on the generated gfx950 ISA (no GPU needed: hipcc cross-compiles)."""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "shadernn_amd", "csrc")


def _device_asm(src, tmp_path):
    out = str(tmp_path / (os.path.basename(src) + ".s"))
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "--cuda-device-only", "-S", src, "-o", out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    return open(out).read()


def test_lds_dma_is_the_only_m0_user(tmp_path):
    """epilogue.h::lds_dma16 writes m0 inside inline assembly the compiler cannot see through.  That is only safe while nothing else
    in those kernels keeps a value in m0 (movrel indexing, s_sendmsg, LDS-direct reads, ds ops that consume m0): every m0 mention in the
    generated code of the translation units that use the DMA must be the `s_mov_b32 m0, sN` of lds_dma16 itself, and each of those must be
    followed by its global_load_lds."""
    users = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip") and "lds_dma16" in open(os.path.join(CSRC, f)).read()]
    assert users, "no translation unit uses lds_dma16 any more: drop this test"
    with ThreadPoolExecutor(max_workers=len(users)) as ex:
        asms = list(ex.map(lambda s: _device_asm(s, tmp_path), users))
    for src, asm in zip(users, asms):
        lines = [l.strip() for l in asm.split("\n")]
        code = [l for l in lines if l and not l.startswith((";", ".", "//")) and not l.endswith(":")]
        m0 = [(i, l) for i, l in enumerate(code) if re.search(r"\bm0\b", l.split(";")[0])]
        assert m0, "%s: lds_dma16 used but no m0 write in the ISA?" % src
        for i, l in m0:
            assert re.fullmatch(r"s_mov_b32 m0, s\d+", l.split(";")[0].strip()), "%s: m0 used outside lds_dma16: %r" % (os.path.basename(src), l)
            nxt = code[i + 1]
            assert nxt.startswith("global_load_lds_dwordx4") or (nxt.startswith("buffer_load_dwordx4") and nxt.split(";")[0].rstrip().endswith(" lds")), \
                "%s: %r is not followed by its DMA but by %r" % (os.path.basename(src), l, nxt)
        assert sum(1 for l in code if l.startswith("global_load_lds") or (l.startswith("buffer_load_dword") and l.split(";")[0].rstrip().endswith(" lds"))) == len(m0)


def test_persistent_wide_kernel_waits_cover_every_load(tmp_path):
    """conv2d_widep_f16.hip counts its s_waitcnt vmcnt(N) by hand (every vector-memory instruction is inline assembly the compiler does not count:
    DESIGN 5.1-12).  tools/audit_vmcnt.py replays the in-order counter over the generated code: no instruction may touch a register an outstanding
    load still writes; the same walk must report hazards once every wait is loosened by one (the check has teeth); and the kernel must not use
    scratch memory (a compiler scratch reload is a vmcnt(0): the prefetch pipeline would run synchronously)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import audit_vmcnt
    asm = _device_asm(os.path.join(CSRC, "conv2d_widep_f16.hip"), tmp_path)
    res = audit_vmcnt.audit(asm, "conv2d_widep_kernel")
    assert len(res) >= 12, "expected the NORM x STATS x FAST instantiations, got %d" % len(res)
    for kernel, loads, depth, hazards, _ in res:
        assert loads >= 150 and depth >= 16, (kernel, loads, depth)
        assert not hazards, "%s:\n%s" % (kernel, "\n".join(hazards[:5]))
    loose = re.sub(r"vmcnt\((\d+)\)", lambda m: "vmcnt(%d)" % (int(m.group(1)) + 1), asm)
    assert all(h for _, _, _, h, _ in audit_vmcnt.audit(loose, "conv2d_widep_kernel"))
    sizes = re.findall(r"conv2d_widep_kernel\S*\.private_seg_size, (\d+)", asm)
    assert sizes and all(int(x) == 0 for x in sizes), sizes
