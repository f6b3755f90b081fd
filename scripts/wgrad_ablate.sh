cd /tmp && export TMPDIR=/tmp
for v in "" a1 a2 a3; do
  if [ -n "$v" ]; then export DIEE_LIB=/root/repo/die-e_amd/libdiee_$v.so; else unset DIEE_LIB; fi
  rocprofv3 --kernel-trace --stats -d /tmp/wgt_$v -o o --output-format csv -- python3 /root/repo/scripts/wgrad_kernel_time.py > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/wgt_$v/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "wgrad" in r["Name"]: print("variant '$v'", r["Name"][:30], r["Calls"], r["AverageNs"])
PY
done
