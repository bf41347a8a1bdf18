"""One stamp for "which kernels is this": sha256 over the sources of libs3shuffle_codec.so (csrc/*.hip, *.inc, *.h, Makefile),
file names included, in sorted order.  bench.py puts it in its JSON line, the profile report (tools/profile_report.py) puts it into
every entry of profiles/traffic_latest.json, and bench.py fills roofline.traffic from that file ONLY when the two agree — a
PMC pass of an older kernel never stands in for the one that was timed.  There is no .git on the GPU box,
so a commit id cannot play this role.
usage: python tools/src_stamp.py"""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_sha256(root: str = ROOT) -> str:
    d = os.path.join(root, "spark-s3-shuffle_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".inc", ".h")) or name == "Makefile":
            h.update(name.encode() + b"\0")
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(kernel_sources_sha256())
