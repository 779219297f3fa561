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
