#!/usr/bin/env python3
"""A/B runner for the GPU box: runs bench.py once per environment variant (same process tree, same box) and prints one compact row
per variant.  Usage: tools/ab.py [--world8] [--so] [--steps K] "NAME=VAL NAME2=VAL2" "..." ...   ("BASE" = no overrides).
Rows: 8-task step ms, dominant-kernel roofline fraction, all-GEMM fraction; with --world8 also the single-task rank
(--emulate-world 8) step; with --so the second-order steps."""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(env_over, extra):
    env = dict(os.environ)
    for kv in env_over.split():
        if "=" in kv:
            k, v = kv.split("=", 1)
            env[k] = v
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-inference", "--no-frontend", "--no-baseline-c2"] + extra
    p = subprocess.run(cmd, env=env, capture_output=True, text=True)
    for ln in reversed(p.stdout.strip().splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    return {"error": (p.stderr or p.stdout)[-400:]}


def main():
    args = sys.argv[1:]
    world8 = "--world8" in args
    so = "--so" in args
    only8 = "--only-world8" in args
    steps = "5"
    if "--steps" in args:
        steps = args[args.index("--steps") + 1]
    variants = [a for a in args if not a.startswith("--") and a != steps]
    for v in variants:
        row = f"== {v}\n"
        base = ["--steps", steps, "--warmup", "2"] + ([] if so else ["--no-second-order"])
        if not only8:
            j = run(v, base)
            if "error" in j:
                row += "  8task ERROR " + j["error"].replace("\n", " | ") + "\n"
            else:
                r = j.get("roofline") or {}
                row += (f"  8task {j['ms_per_step']:.2f} ms  dom {r.get('kernel')} frac {r.get('frac')} ({r.get('launches')} x {r.get('avg_launch_us')} us)  "
                        f"all_gemm {r.get('all_gemm', {}).get('frac')} ({r.get('all_gemm', {}).get('ms_per_meta_step')} ms)")
                if so and j.get("second_order"):
                    row += f"  so {j['second_order']['ms_per_step']:.2f} ms"
                row += "\n"
        if world8 or only8:
            j = run(v, base + ["--emulate-world", "8"])
            if "error" in j:
                row += "  w8 ERROR " + j["error"].replace("\n", " | ") + "\n"
            else:
                r = j.get("roofline") or {}
                row += f"  w8 {j['ms_per_step']:.2f} ms  all_gemm {r.get('all_gemm', {}).get('frac')} ({r.get('all_gemm', {}).get('ms_per_meta_step')} ms)"
                if so and j.get("second_order"):
                    row += f"  so {j['second_order']['ms_per_step']:.2f} ms"
                row += "\n"
        print(row, end="", flush=True)


if __name__ == "__main__":
    main()
