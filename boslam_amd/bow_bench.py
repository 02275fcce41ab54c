"""BASELINE configs[3] on one GPU: L1 score of 1 query against a 100 k-entry x 1000-word database (1.2 GB of
(u32 word, f64 weight) pairs), top-10.  Reports queries/s and the scan kernel's algorithmic HBM rate."""
import time
import numpy as np
from . import bow, synth


def run(ctx, n_entries=100000, repeats=20):
    db = synth.bow_database(n_entries=n_entries)
    d = bow.BowDatabase(ctx)
    d.set_db(db["start"], db["words"], db["vals"])
    d.query(db["q_words"], db["q_vals"], 10)
    ctx.timing(True)
    t0 = time.perf_counter()
    for _ in range(repeats):
        ids, sc = d.query(db["q_words"], db["q_vals"], 10)
    dt = (time.perf_counter() - t0) / repeats
    kt = ctx.timing_read()
    ctx.timing(False)
    scan_ms = kt["k_bow_scan"][0] / kt["k_bow_scan"][1]
    alg = len(db["words"]) * 12.0
    d.close()
    return {"metric": "bow_l1_queries_per_sec", "value": round(1.0 / dt, 1), "unit": "queries/s",
            "workload": f"configs[3] single-GPU shard: 1 query ({len(db['q_words'])} words) vs {n_entries} entries x 1000 words, top-10",
            "scan_kernel_ms": round(scan_ms, 4), "alg_bytes": int(alg),
            "roofline": {"bound": "hbm", "kernel": "k_bow_scan", "achieved": round(alg / (scan_ms * 1e-3) / 1e9, 1), "peak": 8000.0,
                         "unit": "GB/s", "frac": round(alg / (scan_ms * 1e-3) / 1e9 / 8000.0, 4),
                         "note": "algorithmic bytes = 12 B per (word, weight) pair (SURVEY 8d); the kernel streams only the 4-B word ids "
                                 "and touches weights on hits"},
            "top1": int(ids[0]), "top1_score": float(sc[0])}
