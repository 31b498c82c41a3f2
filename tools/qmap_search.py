#!/usr/bin/env python3
"""Which hardware queue should each lane's main / side stream sit on?  Every way of dealing the 2 x <lanes> streams to at most four hardware queues (set partitions as
restricted-growth strings: queue labels do not matter), the variant library libde265_amd/variants/qmap.so (M355_QMAP), one short bench run each; the best ones again, longer.
python tools/qmap_search.py <workload> <lanes> [steps]   ->  a table sorted by ms per picture
QMAP_CHAIN=1: the dependent-chain leg is run too (every picture predicted from the two before it) and printed beside the independent pictures' figure; the second table is
then sorted by the chain's."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHAIN = bool(os.environ.get("QMAP_CHAIN"))
B = ["--no-cpu-baseline", "--no-with-upload", "--no-end-to-end", "--no-cold-refs", "--no-verify"] + ([] if CHAIN else ["--no-dependent-chain"])


def rgs(n, kmax):
    def rec(prefix, m):
        if len(prefix) == n:
            yield tuple(prefix)
            return
        for v in range(min(m + 1, kmax - 1) + 1):
            yield from rec(prefix + [v], max(m, v))
    yield from rec([0], 0)


def run(workload, lanes, qmap, steps):
    env = dict(os.environ, M355_LIB=os.path.join(ROOT, "libde265_amd", "variants", "qmap.so"), M355_QMAP=",".join(map(str, qmap)))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", str(steps), "--warmup", "10", "--pipeline-depth", str(lanes)] + B,
                       env=env, capture_output=True, text=True, timeout=300)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        return d["ms_per_step"], d["ms_per_step_spread"]["p10"], d["ms_per_step_spread"]["p90"], (d.get("dependent_chain") or {}).get("ms_per_step", 0.0)
    except Exception:                                          # noqa: BLE001
        return None


if __name__ == "__main__":
    workload, lanes = sys.argv[1], int(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    res = []
    for q in rgs(2 * lanes, 4):
        t = run(workload, lanes, q, steps)
        if t:
            res.append((t[3] if CHAIN else t[0], q))
            print("%s  %.4f  chain %.4f" % (",".join(map(str, q)), t[0], t[3]), flush=True)
    res.sort()
    print("== the best 12 again, %d steps, twice" % (4 * steps))
    for _, q in res[:12]:
        a, b = run(workload, lanes, q, 4 * steps), run(workload, lanes, q, 4 * steps)
        a, b = a or (0, 0, 0, 0), b or (0, 0, 0, 0)
        print("%s  %.4f (p10 %.4f p90 %.4f) chain %.4f |  %.4f (p10 %.4f p90 %.4f) chain %.4f" % ((",".join(map(str, q)),) + tuple(a) + tuple(b)), flush=True)
    print("== the worst 3: " + "; ".join("%s %.4f" % (",".join(map(str, q)), t) for t, q in res[-3:]))
