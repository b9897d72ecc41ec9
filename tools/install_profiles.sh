#!/bin/bash
# Run HERE after `git rev-parse HEAD > gpurun_out/profiles_r6_commit.txt; gpurun -- bash tools/collect_profiles.sh`:
# earlier rounds' files go to profiles/history/, the freshly collected set (gpurun_out/profiles_$RT/) becomes profiles/.
set -e
cd "$(dirname "$0")/.."
mkdir -p profiles/history
RT=${NSK_ROUND_TAG:-r6}
for f in profiles/r2_* profiles/r3_* profiles/r4_* profiles/r5_*; do [ -e "$f" ] && git mv -f "$f" profiles/history/ 2>/dev/null || true; done
# (NSK_PROFILE_PARTIAL=1: a partial collection replaces only the files it produced)
[ -z "$NSK_PROFILE_PARTIAL" ] && for f in profiles/${RT}_*; do [ -e "$f" ] && rm -f "$f"; done
cp gpurun_out/profiles_$RT/${RT}_* profiles/
[ -e gpurun_out/profiles_$RT/traffic.json ] && cp gpurun_out/profiles_$RT/traffic.json profiles/traffic.json
[ -e gpurun_out/profiles_$RT/issue.json ] && cp gpurun_out/profiles_$RT/issue.json profiles/issue.json
for f in gpurun_out/profiles_$RT/config4_shards_*.json gpurun_out/profiles_$RT/config5_shards_*.json; do [ -e "$f" ] && cp "$f" profiles/${RT}_$(basename $f); done
if [ -z "$NSK_PROFILE_PARTIAL" ]; then cp gpurun_out/profiles_${RT}_commit.txt profiles/${RT}_COMMIT.txt
else cat gpurun_out/profiles_${RT}_commit.txt >> profiles/${RT}_COMMIT.txt; fi
# the library the collection ran (bench.py compares it with the one that runs: traffic_stale)
python3 -c "import sys; sys.path.insert(0, '.'); import bench; from numbskull_amd import _lib; print('library:', bench.library_fingerprint(_lib))" >> profiles/${RT}_COMMIT.txt
rm -f profiles/*.err
ls profiles
