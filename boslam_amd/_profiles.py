"""Committed profiler summaries (profiles/*.json) as inputs of bench.py: hardware counters per launch are a property of a kernel and its
workload, so bench.py combines the committed counters with kernel times it measures live.  That is only sound while the kernels are the ones
the counters were collected on: every summary carries `_meta` = {commit, sources: {file: sha16 of the kernel source}}, written by
the tools that produce it ON THE BOX from the sources that were running (tools/pmc_summary.py, tools/ba_pmc.py, tools/pmc_traffic.py), and a
block derived from a summary whose source hashes differ from the running build's is reported as stale instead of as a current fraction."""
import hashlib
import json
import os

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CSRC = os.path.join(_ROOT, "boslam_amd", "csrc")


def source_sha16(name):
    try:
        with open(os.path.join(_CSRC, name), "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        return None


def meta(sources, commit=None):
    """The `_meta` block a summarising tool writes next to its counters."""
    return {"commit": commit or os.environ.get("BH_COMMIT") or None, "sources": {s: source_sha16(s) for s in sources}}


def load(rel, sources):
    """(data or None, stale reason or None).  rel: path under the repo root; sources: the kernel files the derived figures depend on."""
    try:
        with open(os.path.join(_ROOT, rel)) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None, "absent"
    m = d.get("_meta") or {}
    have = m.get("sources") or {}
    for s in sources:
        cur = source_sha16(s)
        if cur is None or have.get(s) != cur:
            return d, f"{rel} was collected on another build of {s} (summary {have.get(s)}, running {cur}, commit {m.get('commit')})"
    return d, None


def stale_block(rel, why):
    return {"source": rel, "stale": True, "note": "counters not combined with the live times: " + why}
