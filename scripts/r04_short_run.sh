#!/bin/bash
# Where a short run() goes, round 4 (GPU box only): host phases, device time line, carried against built-on-entry
export TMPDIR=/tmp
R=$PWD
out=$R/gpurun_out/r04_short
mkdir -p "$out"
python3 scripts/run_phases.py 20 c3 0 > "$out/phases_build.log" 2>&1
python3 scripts/run_phases.py 20 c3 1 > "$out/phases_carry.log" 2>&1
python3 scripts/carry_probe.py c3 20 100 256 > "$out/carry_probe.log" 2>&1
rm -rf /tmp/srt_b /tmp/srt_c /tmp/rbt
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/srt_b -- python3 "$R/scripts/short_run_timeline.py" run c3 20 > "$out/timeline_build.log" 2>&1)
python3 scripts/short_run_timeline.py report /tmp/srt_b >> "$out/timeline_build.log" 2>&1
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/srt_c -- python3 "$R/scripts/short_run_timeline.py" run c3 20 carry > "$out/timeline_carry.log" 2>&1)
python3 scripts/short_run_timeline.py report /tmp/srt_c >> "$out/timeline_carry.log" 2>&1
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/rbt -- python3 "$R/scripts/rebuild_timeline.py" run > "$out/rebuild_timeline.log" 2>&1)
python3 scripts/rebuild_timeline.py report /tmp/rbt >> "$out/rebuild_timeline.log" 2>&1
python3 scripts/noop_probe.py > "$out/noop.log" 2>&1
for f in phases_build phases_carry carry_probe timeline_build timeline_carry rebuild_timeline noop; do echo "---- $f"; grep -v "amdgpu.ids\|^W2\|^E2\|rocprofv3" "$out/$f.log" | tail -32; done
