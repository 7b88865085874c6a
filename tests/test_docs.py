"""Documentation lint (CPU): every environment switch the native code reads is listed in DESIGN.md, and every file the README's layout table
names exists -- the switches are how the measurements in DESIGN.md are reproduced."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read(path):
    with open(os.path.join(ROOT, path), encoding="utf-8") as f:
        return f.read()


def test_every_environment_switch_is_documented():
    design = _read("DESIGN.md")
    names = set()
    for pat in ("shadernn_amd/csrc/*.hip", "shadernn_amd/csrc/*.h", "shadernn_amd/host/*.cpp", "shadernn_amd/host/*/*.cpp", "shadernn_amd/*.py"):
        for path in glob.glob(os.path.join(ROOT, pat)):
            with open(path, encoding="utf-8") as f:
                src = f.read()
            names.update(re.findall(r'getenv\("(SNN[A-Z0-9_]+)"\)', src))
            names.update(re.findall(r'option\("(SNN[A-Z0-9_]+)"\)', src))  # snnhip::option(): the options registry (override, else environment)
            names.update(re.findall(r'environ(?:\.get)?\(\s*"(SNN[A-Z0-9_]+)"', src))
    assert names, "no switches found: the scan is broken"
    missing = sorted(n for n in names if n not in design)
    assert not missing, "environment switches read by the code but absent from DESIGN.md: %s" % missing


def test_readme_layout_paths_exist():
    readme = _read("README.md")
    for path in re.findall(r"^\| `([^`]+)` \|", readme, flags=re.M):
        path = path.rstrip("/")
        if "*" in path:
            assert glob.glob(os.path.join(ROOT, path)), path
        else:
            assert os.path.exists(os.path.join(ROOT, path)), path


def test_committed_pmc_passes_were_taken_with_these_kernel_sources():
    """bench.py fills `roofline.traffic` from profiles/pmc_latest.json only when the file's kernel-source fingerprint is the build's (shadernn_amd/fingerprint.py);
    a kernel edit without a fresh `tools/gpurun.sh <t> <tag> prof:c1 ... prof:c5` + `tools/collect_profiles.py` would put `traffic: null` on the driver's line."""
    import json
    import sys

    sys.path.insert(0, ROOT)
    from shadernn_amd import fingerprint

    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
    assert pmc, "profiles/pmc_latest.json is empty"
    shas = {v.get("csrc_sha16") for v in pmc.values()}
    assert shas == {fingerprint.csrc_sha16()}, "profiles/pmc_latest.json was taken with kernel sources %s, the tree is %s: re-run the profile pass" % (sorted(shas), fingerprint.csrc_sha16())
    # the dominant kernels of the five bench configs have a traffic record
    for name in ("conv_kxk_c1o16_wino3x3_c16o16_kernel<5,16,2,2>", "conv2d_wino_kernel<true,1,1>", "conv2d_wino_kernel<true,1,2>", "conv2d_ksplit_pair_kernel<4>", "irb_image_kernel<6,6,13,true,true>", "conv2d_stem32_kernel<3,1,2,true>"):
        assert name in pmc and pmc[name].get("hbm_bytes_per_launch", 0) > 0, name


def test_rocprof_summaries_of_this_round_carry_the_steady_state_table():
    for c in ("c1_conv3x3", "c2", "c3_resnet18", "c4_mobilenetv2", "c5_candy_fp16"):
        text = _read("profiles/r06_%s_rocprofv3_summary.md" % c)
        assert "## kernel stats (--kernel-trace --stats)" in text and "## steady state (same trace, per-dispatch durations in launch order)" in text, c
