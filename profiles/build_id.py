#!/usr/bin/env python3
"""Build id of the library as the Makefile computes it (csrc/Makefile, BUILD_ID): sha256 of the sources and headers, in the
Makefile's order, first 12 hex digits.  `python3 profiles/build_id.py` prints it; tests/test_host_logic.py compares it with
xvec_version() of the built library and with the `build` fields of profiles/traffic.json."""
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "speaker-recognition-x-vectors_amd", "csrc")


def source_build_id() -> str:
    mk = open(os.path.join(CSRC, "Makefile")).read()
    srcs = re.search(r"^SRCS = (.*)$", mk, re.M).group(1).split()
    hdrs = re.search(r"^HDRS = (.*)$", mk, re.M).group(1).split()
    h = hashlib.sha256()
    for f in srcs + hdrs:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:12]


if __name__ == "__main__":
    print(source_build_id())
