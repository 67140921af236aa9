set -e
out=gpurun_out/full1
mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $out/tests.txt 2>&1 || { tail -40 $out/tests.txt; exit 1; }
tail -2 $out/tests.txt
timeout -k 10 150 python scripts/soak.py 1 100000 > $out/s1.txt 2>&1
timeout -k 10 150 python scripts/soak.py 0 100000 > $out/s2.txt 2>&1
timeout -k 10 150 python scripts/soak.py 0 100000 > $out/s3.txt 2>&1
cat $out/s?.txt
