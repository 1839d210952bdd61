#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY.  Copies the reference's own tests for the hot path, byte for byte, from the reference checkout
into the git-ignored tests/_reference_tests/ (reference sources never enter this repository's history; the directory travels
to the GPU box with the working tree like the built .so files).  tests/test_gpu_reference_suite.py runs them against the
product.  Called by __graft_entry__.build() where /root/reference exists."""
import hashlib
import os
import shutil
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REF = os.environ.get("COLPALI_REFERENCE", "/root/reference")
FILES = ("tests/utils/test_processing_utils.py", "tests/loss/test_li_losses.py")
# the model files whose forward colpali_amd.patch_colpali_engine(models=True) wraps: the GPU tests and bench.py's "VLM in the loop"
# leg run the REAL classes (random init), so these travel the same way -- verbatim, git-ignored, under tests/_reference_pkg/
MODEL_FILES = ("colpali_engine/loss/late_interaction_losses.py",         # bench.py: the reference's own loss module timed beside ours
               "colpali_engine/utils/processing_utils.py",              # bench.py: cpu_baseline.kind = "reference"
               "colpali_engine/utils/torch_utils.py",
               "colpali_engine/models/paligemma/colpali/modeling_colpali.py",
               "colpali_engine/models/qwen2/colqwen2/modeling_colqwen2.py",
               "colpali_engine/models/qwen2_5/colqwen2_5/modeling_colqwen2_5.py",
               "colpali_engine/models/idefics3/colidefics3/modeling_colidefics3.py")


def main() -> int:
    if not os.path.isdir(REF):
        print(f"fetch_reference_tests: {REF} not present (GPU box): keeping what the working tree carries")
        return 0
    dst_dir = os.path.join(ROOT, "tests", "_reference_tests")
    os.makedirs(dst_dir, exist_ok=True)
    lines = []
    for rel in FILES:
        src = os.path.join(REF, rel)
        dst = os.path.join(dst_dir, os.path.basename(rel))
        shutil.copyfile(src, dst)
        lines.append(f"{hashlib.sha256(open(dst, 'rb').read()).hexdigest()}  {rel}")
    pkg_dir = os.path.join(ROOT, "tests", "_reference_pkg")
    for rel in MODEL_FILES:
        src = os.path.join(REF, rel)
        if not os.path.exists(src):
            continue
        dst = os.path.join(pkg_dir, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(src, dst)
        lines.append(f"{hashlib.sha256(open(dst, 'rb').read()).hexdigest()}  {rel}")
    with open(os.path.join(dst_dir, "SOURCES.txt"), "w") as f:
        f.write("# verbatim copies from the reference checkout (sha256, path relative to it); not committed\n" + "\n".join(lines) + "\n")
    print("fetch_reference_tests:", ", ".join(os.path.basename(r) for r in FILES), "->", dst_dir)
    return 0


if __name__ == "__main__":
    sys.exit(main())
