#!/bin/bash
# round 4, session 28: bench stage of the re-collection at the round's last library -- every bench line, the
# two-rank lines, then the whole GPU suite (it contains the 8-shard runs) and the smoke run
export NSK_PROFILE_PARTIAL=1 NSK_PROFILE_STAGE=bench NSK_PROFILE_FULL_TESTS=1
bash tools/collect_profiles.sh
