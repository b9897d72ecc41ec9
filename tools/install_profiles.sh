#!/bin/bash
# Run HERE after `gpurun -- bash tools/collect_profiles.sh`: profiles/r2_* go to profiles/history/, the
# freshly collected r3 set (gpurun_out/profiles_r3/) becomes profiles/.
set -e
cd "$(dirname "$0")/.."
mkdir -p profiles/history
for f in profiles/r2_*; do [ -e "$f" ] && git mv -f "$f" profiles/history/ 2>/dev/null || true; done
# (NSK_PROFILE_PARTIAL=1: a partial collection replaces only the files it produced)
[ -z "$NSK_PROFILE_PARTIAL" ] && for f in profiles/r3_*; do [ -e "$f" ] && rm -f "$f"; done
cp gpurun_out/profiles_r3/r3_* profiles/
cp gpurun_out/profiles_r3/traffic.json profiles/traffic.json
for f in gpurun_out/profiles_r3/config4_shards_*.json; do [ -e "$f" ] && cp "$f" profiles/r3_$(basename $f); done
if [ -z "$NSK_PROFILE_PARTIAL" ]; then cp gpurun_out/profiles_r3_commit.txt profiles/r3_COMMIT.txt
else cat gpurun_out/profiles_r3_commit.txt >> profiles/r3_COMMIT.txt; fi
rm -f profiles/*.err
ls profiles
