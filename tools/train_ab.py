#!/usr/bin/env python
"""Same-box A/B of the configs[2] train step (16 clips x 2 s, predictor heads on) under environment switches: each variant runs
in its own process (the switches are read at import), `--steps` timed steps after `--warmup`, alternating A B A B.

    python tools/train_ab.py "FAC_EXP_STALE_PACKS=1" ["FAC_DISC_STREAMS=1 FAC_PRED_STREAMS=1" ...]   (baseline = no switch)
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def run(env_str, steps, warmup):
    env = dict(os.environ)
    for kv in env_str.split():
        k, v = kv.split("=", 1)
        env[k] = v
    out = subprocess.run([sys.executable, os.path.join(HERE, "train_bench.py"), "--predictors", "--steps", str(steps), "--warmup", str(warmup)],
                         env=env, capture_output=True, text=True)
    for line in out.stdout.splitlines():
        if line.startswith("{"):
            return json.loads(line)["ms_per_step"]
    raise SystemExit(out.stdout + out.stderr)


def main():
    variants = [""] + [a for a in sys.argv[1:] if "=" in a]
    res = {v: [] for v in variants}
    for rep in range(2):
        for v in variants:
            res[v].append(run(v, 6, 2))
            print(rep, repr(v), res[v][-1], flush=True)
    print(json.dumps({(v or "baseline"): r for v, r in res.items()}))


if __name__ == "__main__":
    main()
