#!/bin/bash
# round 4, session 29: one accumulator copy beyond 2^18 weights (the 50M LR graph and its shards): the light
# profile and the bench line of lr50m_learn, lr5m_learn as the control, then the whole GPU suite and smoke
export NSK_PROFILE_PARTIAL=1 NSK_PROFILE_WORKLOADS=" " NSK_PROFILE_LIGHT_WORKLOADS="lr50m_learn" NSK_PROFILE_SKIP_DEFAULT=1
export NSK_PROFILE_BENCH_WORKLOADS="lr5m_learn" NSK_PROFILE_BENCH_ONLY="lr50m_learn" NSK_PROFILE_FULL_TESTS=1
bash tools/collect_profiles.sh
