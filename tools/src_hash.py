"""sha256 over the sources libgsplat_hip.so is built from and the host code the GPU tests drive it through (sorted paths,
path + NUL + bytes): what `profiles/r06_gpu_tests.log` was run on.  tests/test_profiles.py compares the log's value with the tree."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATTERNS = ("vk3dgaussiansplatting_amd/csrc/*.hip", "vk3dgaussiansplatting_amd/csrc/*.cpp", "vk3dgaussiansplatting_amd/csrc/*.h",
            "vk3dgaussiansplatting_amd/csrc/Makefile", "vk3dgaussiansplatting_amd/*.py", "include/*.h", "include/*.hpp",
            "tools/gsplat_bench.cpp", "oracle/gs_oracle.c", "oracle/__init__.py")


def source_hash(root=ROOT):
    h = hashlib.sha256()
    files = sorted(p for pat in PATTERNS for p in glob.glob(os.path.join(root, pat)))
    for p in files:
        h.update(os.path.relpath(p, root).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


if __name__ == "__main__":
    print(source_hash())
