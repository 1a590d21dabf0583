#!/bin/bash
# Build and run the calibration program ON THE GPU BOX from the repo root: bash scripts/calibrate/run.sh <tag>
# -> gpurun_out/calib_<tag>/{peaks.json, pmc_fetch.md, pmc_write.md}; copy what should be judged into profiles/.
set -e
TAG=${1:-r03}
OUT=gpurun_out/calib_${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/calibrate/calib.hip -o $OUT/calib
$OUT/calib > $OUT/peaks.json
cat $OUT/peaks.json
for pass in fetch:FETCH_SIZE write:WRITE_SIZE; do
  name=${pass%%:*}; ctr=${pass#*:}
  rocprofv3 --kernel-trace --pmc $ctr --output-format rocpd -d $OUT/pmc_$name -o $name -- $OUT/calib pmc > $OUT/pmc_$name.log 2>&1
  python3 scripts/rocpd_pmc_summary.py $(find $OUT/pmc_$name -name "*.db" | head -1) $OUT/pmc_$name.md
done
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/calib
