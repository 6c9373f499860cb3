#!/usr/bin/env python3
"""profiles/PROVENANCE.json: content hashes (sha256, 16 hex digits) of the files a bench line's cited measurements depend
on, with the commit that last touched each -- written in the build container (git is available here, not on the GPU box)
and COMMITTED, so that `roofline.traffic_source` can say on any box whether the counter pass it cites was taken on the
kernel source that is running.  Run it after changing a listed file and before committing."""
import hashlib, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["cellmapper_amd/csrc/knn_phase1_q4.h", "cellmapper_amd/csrc/knn_phase1_f16.h", "cellmapper_amd/csrc/knn_phase1_r16.h",
         "cellmapper_amd/csrc/knn_kernels.hip", "cellmapper_amd/csrc/graph_kernels.hip", "profiles/phase1_hbm_traffic.json"]


def sha16(path):
    with open(os.path.join(ROOT, path), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def last_commit(path):
    try:
        return subprocess.check_output(["git", "log", "-1", "--format=%h", "--", path], cwd=ROOT, text=True).strip() or None
    except Exception:
        return None


def main():
    out = {"note": "sha256[:16] of each file as committed; `commit` = the last commit that touched it when this file was "
                   "written (a change committed together with this file shows the previous commit: the hash is what counts)",
           "files": {p: {"sha256_16": sha16(p), "commit": last_commit(p)} for p in FILES}}
    with open(os.path.join(ROOT, "profiles", "PROVENANCE.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out["files"], indent=1))


if __name__ == "__main__":
    main()
