#!/bin/bash
# Where do the activation (X) reads of the stream kernel come from?  FETCH_SIZE (memory side) and L2 hit / miss counts per launch, 16 and 48 rows.
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O=gpurun_out/s20; mkdir -p $O
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum -d $R/$O/pmc -o pmc -- python $R/tools/bench_ttft.py --ns 16,48 --reps 2 > $R/$O/pmc.log 2>&1 )
db=$(find $O/pmc -name "*.db" | head -1); [ -n "$db" ] && python tools/pmc_dump.py $db k_stream > $O/pmc_stream.txt 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_READ_sum -d $R/$O/pmc2 -o pmc2 -- python $R/tools/bench_ttft.py --ns 16,48 --reps 2 > $R/$O/pmc2.log 2>&1 )
db=$(find $O/pmc2 -name "*.db" | head -1); [ -n "$db" ] && python tools/pmc_dump.py $db k_stream > $O/pmc2_stream.txt 2>&1
find $O -name "*.db" -delete
cat $O/pmc_stream.txt $O/pmc2_stream.txt | head -120; tail -3 $O/pmc.log $O/pmc2.log
