export TMPDIR=/tmp
for rep in 1 2; do for p in 50 55 60 67; do
NBLS_HALVES_SPLIT_PCT=$p python tools/exp_time.py 65536 8 2>&1 | tail -1 | cut -c1-60 | sed "s/^/split=$p /"
NBLS_HALVES_SPLIT_PCT=$p python tools/exp_time.py 16384 20 2>&1 | tail -1 | cut -c1-60 | sed "s/^/split=$p /"
done; done
