#!/bin/bash
# which HIP API call stalls?  (hip-trace only, no counters)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/hiptrace
rocprofv3 --hip-trace --output-format csv -d gpurun_out/hiptrace -- python3 tools/step_times.py > gpurun_out/hiptrace.log 2>&1
f=$(find gpurun_out/hiptrace -name "*hip_api_trace.csv" | head -1)
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t_end=int(rows[-1]["End_Timestamp"])
# last 400 ms
idx=[i for i,r in enumerate(rows) if int(r["End_Timestamp"])-int(r["Start_Timestamp"])>5e6 and t_end-int(r["Start_Timestamp"])<200e6]
for i in idx:
    for r in rows[max(0,i-6):i+3]:
        print(f'{(t_end-int(r["Start_Timestamp"]))/1e6:9.3f}ms before end  dur {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6:8.3f} ms  {r["Function"]}')
    print("----")
PY
grep "step" gpurun_out/hiptrace.log | tr "\n" " " | cut -c1-400
