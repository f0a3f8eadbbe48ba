#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference's OWN CPU-oracle loops.

Run in the build container (needs /root/reference): the loops of
/root/reference/main.mm:128-159 (non-causal) and :550-578 (causal) are compiled
by oracle/build_ref.sh into oracle/_ref/libfa_ref_slices.so and executed here;
the arrays they return are the golden vectors. Inputs come from the reference's
initRandom (main.mm:24-30, seed 42, so Q == K == V as in the reference) and,
for the independent-Q/K/V cases, from our restatement of the same generator
with seeds 42/43/44 (validated against the reference generator for seed 42).

Only data is written: inputs are regenerated from the seed, expected outputs are
stored as fp32 arrays.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle  # noqa: E402

D = 64


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main() -> None:
    oracle.build()
    assert oracle.have_ref(), "oracle/_ref not built (needs /root/reference)"
    out = {}
    meta = {"D": D, "scale": float(oracle.ref().ref_scale()), "cases": {}}

    # the reference's generator itself: first values + digest of 1024*64 draws
    r = oracle.ref_init_random(1024 * D)
    out["init_random_seed42_head"] = r[:256].copy()
    meta["init_random_seed42_sha256_65536"] = sha(r)

    # reference mode: Q == K == V (main.mm:117-119)
    for n in (128, 256):
        x = oracle.ref_init_random(n * D).reshape(n, D)
        out[f"noncausal_same_n{n}"] = oracle.ref_noncausal(x, x, x)
        out[f"causal_same_n{n}"] = oracle.ref_causal(x, x, x)
        meta["cases"][f"same_n{n}"] = {"seed": [42, 42, 42], "input_sha256": sha(x)}

    # independent Q, K, V (SURVEY.md section 4 weakness 1), incl. a ragged N
    for n in (128, 200):
        q = oracle.init_random(n * D, 42).reshape(n, D)
        k = oracle.init_random(n * D, 43).reshape(n, D)
        v = oracle.init_random(n * D, 44).reshape(n, D)
        out[f"noncausal_indep_n{n}"] = oracle.ref_noncausal(q, k, v)
        out[f"causal_indep_n{n}"] = oracle.ref_causal(q, k, v)
        meta["cases"][f"indep_n{n}"] = {"seed": [42, 43, 44],
                                         "input_sha256": [sha(q), sha(k), sha(v)]}

    # N=1024 non-causal, the size the reference itself verifies (main.mm:11):
    # the full tensor is 256 KiB, keep every 16th row (16 KiB) + the probes
    x = r.reshape(1024, D)
    o = oracle.ref_noncausal(x, x, x)
    out["noncausal_same_n1024_rows_step16"] = o[::16].copy()
    meta["noncausal_same_n1024_sha256"] = sha(o)
    meta["probes_n1024"] = {"O[0]": float(o.flat[0]), "O[1]": float(o.flat[1]),
                            "O[last]": float(o.flat[-1])}

    np.savez_compressed(os.path.join(HERE, "reference_cpu_oracle.npz"), **out)
    with open(os.path.join(HERE, "reference_cpu_oracle.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", sorted(out), file=sys.stderr)


if __name__ == "__main__":
    main()
