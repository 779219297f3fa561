"""Identity of the kernel sources a measurement belongs to."""
import glob
import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def csrc_sha():
    """One hash over csrc/*.hip + common.h (the git blob hash of each file, like `git hash-object`): profiles record it, and
    bench.py prints a profile's numbers only when the sources it runs on still hash to the same value."""
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(_CSRC, "*.hip")) + [os.path.join(_CSRC, "common.h")]):
        data = open(f, "rb").read()
        h.update(hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest().encode())
    return h.hexdigest()


def traffic_profile(suffix=""):
    """The newest committed PMC traffic profile (profiles/rNN_traffic<suffix>.json: tools/pmc_traffic.py) measured on THESE kernel
    sources, or None: a profile is only valid for the csrc hash it records."""
    import json
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    for f in sorted(glob.glob(os.path.join(root, f"r[0-9][0-9]_traffic{suffix}.json")), reverse=True):
        try:
            tj = json.load(open(f))
        except Exception:          # noqa: BLE001 -- an unreadable profile is no profile
            continue
        if tj.get("csrc_sha") == csrc_sha():
            return tj
    return None
