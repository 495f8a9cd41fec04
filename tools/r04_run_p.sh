#!/bin/bash
# final evidence of the round: profile (traces, counters, timers), then the round-end checks, on one box
bash tools/profile_round4.sh > gpurun_out/profile_round4.log 2>&1
bash tools/round_end4.sh
