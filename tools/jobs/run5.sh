for cf in 2 3 4 6; do
echo "cell factor $cf"
SCCD_CELL_FACTOR=$cf timeout 300 python bench.py --workload boxes1m --steps 50 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config']['candidates'], d['roofline']['class_ms_per_step'])"
SCCD_CELL_FACTOR=$cf SCCD_OVERLAP=0 timeout 300 python bench.py --steps 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config']['candidates_per_step'], d['roofline']['class_ms_per_step'])"
SCCD_CELL_FACTOR=$cf timeout 300 python bench.py --steps 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['class_ms_per_step'])"
done
bash tools/jobs/pmc_quick.sh cloth1m sweep_band_k
