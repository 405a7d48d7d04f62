"""bench.py against a variant library (tools/build_variant.sh):  python tools/bench_variant.py path/to/libfsvit_X.so [bench args...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fewshot_vit_amd import _lib            # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import bench                                # noqa: E402

sys.exit(bench.main(sys.argv[2:]))
