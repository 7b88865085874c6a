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
