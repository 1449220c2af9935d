#!/bin/bash
# round 6, GPU session 2: coopStitch by kind of join (profile build)
cd ${GRAFT_REPO_ROOT:-.}
V=star_amd/lib/variants
bash tools/session.sh ab r06s2 "new|-|" "prof|$V/libstaramd_prof.so|"
grep "coopStitch by kind\|profile (k" gpurun_out/r06s2/ab.txt | head -4
