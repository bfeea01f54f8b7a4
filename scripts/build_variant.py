"""Build a VARIANT of libcnmfe_hip.so from the committed tree plus patches, for A/B runs on one GPU lease.

    python scripts/build_variant.py NAME [patch ...]

exports HEAD (plus the uncommitted changes of the working tree) into /tmp/cnmfe_var_NAME, applies the patches there, compiles for gfx950 and copies the
library to cnmf_e_amd/variants/libcnmfe_NAME.so (git-ignored, shipped by gpurun).  `CNMFE_LIB=<that path>` makes cnmf_e_amd/_lib.py load it instead of
the in-tree library, `CNMFE_OPTS="name=value,..."` presets cnmfe_set_option tunables -- so the test suite and bench.py run unmodified on either build.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    name, patches = sys.argv[1], [("R:" if p.startswith("R:") else "") + os.path.abspath(p[2:] if p.startswith("R:") else p) for p in sys.argv[2:]]     # "R:<patch>": applied in reverse
    work = "/tmp/cnmfe_var_" + name
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    for sub in ("cnmf_e_amd/csrc", "include"):
        shutil.copytree(os.path.join(ROOT, sub), os.path.join(work, sub))
    for f in ("cnmf_e_amd/build.py", "cnmf_e_amd/__init__.py"):
        shutil.copy(os.path.join(ROOT, f), os.path.join(work, f))
    subprocess.run(["git", "init", "-q"], cwd=work, check=True)
    for p in patches:
        rev = p.startswith("R:")
        p = p[2:] if rev else p
        subprocess.run(["git", "apply", "--whitespace=nowarn"] + (["-R"] if rev else []) + [p], cwd=work, check=True)
        print("applied" + (" in reverse" if rev else ""), os.path.relpath(p, ROOT), flush=True)
    subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, '.'); from cnmf_e_amd import build; build.build(force=True, verbose=False)"], cwd=work, check=True)
    dst = os.path.join(ROOT, "cnmf_e_amd", "variants")
    os.makedirs(dst, exist_ok=True)
    out = os.path.join(dst, "libcnmfe_%s.so" % name)
    shutil.copy(os.path.join(work, "cnmf_e_amd", "libcnmfe_hip.so"), out)
    print("built", os.path.relpath(out, ROOT))


if __name__ == "__main__":
    main()
