#!/usr/bin/env python3
"""ISA guard of libeffconf.so (no hazardous packed-fp32 forms in product kernels): thin front end of efficientconformer_amd/_isa_guard.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientconformer_amd._isa_guard import main, scan, demangle, check, EXEMPT, BRANCH_FREE  # noqa: E402,F401

if __name__ == "__main__":
    sys.exit(main())
