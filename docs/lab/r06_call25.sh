#!/bin/bash
# round 6, call 25: the round's profile set of the driver's commands (kernel-trace stats, timelines, PMC passes)
bash tools/profile_all.sh r06
