"""Parity ledger: every end-to-end GPU parity test records the error it ACHIEVED next to the bound it asserts.

`check(test, key, achieved, bound)` asserts `achieved <= bound` and remembers both; tests/conftest.py writes the collected rows to
``$UDM_LEDGER`` (default ``gpurun_out/parity_ledger.json``) at the end of a GPU session.  The file committed under ``profiles/`` is a copy of
that output; bounds in the tests are set to <= 3x the achieved error recorded there (north_star: 1e-3-class relative error on bf16 loss / logits,
bit-exact masks).
"""
import json
import os

ROWS = []


def record(test, key, achieved, bound=None, note=None):
    row = dict(test=test, key=key, achieved=float(achieved), bound=None if bound is None else float(bound))
    if note:
        row["note"] = note
    ROWS.append(row)
    return row


def check(test, key, achieved, bound, note=None):
    record(test, key, achieved, bound, note)
    assert float(achieved) <= float(bound), f"{test}: {key} = {float(achieved):.3e} exceeds the stated bound {float(bound):.3e}"


def dump(path=None):
    if not ROWS:
        return None
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = path or os.environ.get("UDM_LEDGER") or os.path.join(root, "gpurun_out", "parity_ledger.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(dict(rows=ROWS), f, indent=1)
    return path
