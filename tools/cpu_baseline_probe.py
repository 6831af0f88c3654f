#!/usr/bin/env python3
"""CPU baseline on the GPU box's host cores: the 8-process concurrent leg of bench.py under allocator / OpenMP-runtime settings (VERDICT r05 item 9:
pinned processes, one inter-op thread, MALLOC_ARENA_MAX, passive OpenMP waiting).  One line per variant: seconds per 8-task meta-step."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

MALLOC = {"MALLOC_ARENA_MAX": "1", "MALLOC_MMAP_THRESHOLD_": "33554432", "MALLOC_TRIM_THRESHOLD_": "4294967295", "MALLOC_TOP_PAD_": "268435456"}
PASSIVE = {"OMP_WAIT_POLICY": "PASSIVE", "GOMP_SPINCOUNT": "0"}
BOTH = dict(MALLOC, **PASSIVE)
VARIANTS = [("pinned x16", [16], True, {}, 0), ("pinned x16 + passive", [16], True, PASSIVE, 0), ("pinned x16 + passive + malloc", [16], True, BOTH, 0),
            ("pinned x8 + passive + malloc", [8], True, BOTH, 0), ("pinned x4 + passive + malloc", [4], True, BOTH, 0),
            ("pinned x32 + passive + malloc", [32], True, BOTH, 0), ("unpinned x2", [2], False, {}, 0), ("unpinned x2 + passive + malloc", [2], False, BOTH, 0),
            ("unpinned x4 + passive + malloc", [4], False, BOTH, 0)]


def main():
    only = sys.argv[1:]
    for name, legs, pin, env, interop in VARIANTS:
        if only and not any(o == name for o in only):
            continue
        try:
            r = bench.cpu_baseline_concurrent(legs, pin=pin, env=env, interop=interop)[0]
            print(json.dumps({"variant": name, "s_per_meta_step": r["s_per_meta_step"], "slowest_task_s": r["slowest_task_s"], "fastest_task_s": r["fastest_task_s"],
                              "cores": r["cores"]}), flush=True)
        except Exception as ex:  # noqa: BLE001
            print(json.dumps({"variant": name, "error": str(ex)[:200]}), flush=True)


if __name__ == "__main__":
    main()
