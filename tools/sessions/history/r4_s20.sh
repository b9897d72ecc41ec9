#!/bin/bash
# round 4, session 20: re-collection after the last kernel changes -- the learning workloads' profiles, every bench
# line, the two-rank lines and the 8-shard runs (NSK_PROFILE_PARTIAL: the other workloads' profiles stay)
export NSK_PROFILE_PARTIAL=1 NSK_PROFILE_WORKLOADS="lr5m_learn boolw4m_learn" NSK_PROFILE_LIGHT_WORKLOADS="lr50m_learn"
bash tools/collect_profiles.sh
