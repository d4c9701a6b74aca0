cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5e
python - > gpurun_out/r5e/extract.txt 2>&1 <<'PY'
import bench, json, time
for gl in (600_000, 2_400_000):
    t = time.time()
    r = bench.pipeline_extract_leg(genome_len=gl, threads=(1, 8, 16))
    print(gl, json.dumps({k: v for k, v in r.items() if k != 'note'}), round(time.time() - t, 1))
PY
