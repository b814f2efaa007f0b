#!/usr/bin/env python3
"""usage: python scripts/check_exec_prologue.py [extra hipcc flags...]     exit 1 if a kernel has a hit
(cam_nor_physics_amd/isa_check.py: vector instructions the backend left in front of a loop-exit block's EXEC restore)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cam_nor_physics_amd import isa_check

hits = isa_check.check_sources(sys.argv[1:])
if hits:
    print(isa_check.format_hits(hits))
print(f"check_exec_prologue: {len(hits)} hit(s)")
sys.exit(1 if hits else 0)
