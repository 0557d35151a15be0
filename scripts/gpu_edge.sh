# edge-shaped inputs through the CLI: one 50 Mbp contig; 200 000 short contigs through the padded short-contig pass;
# an all-N contig next to normal ones; timing + sanity of the TSVs
cd $GRAFT_REPO_ROOT
python - <<'PY'
import numpy as np, sys
from pathlib import Path
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import make_model_dir
make_model_dir(Path('/tmp/model_brain'))
rng = np.random.Generator(np.random.PCG64(9))
A = np.frombuffer(b"ACGT", np.uint8)
def wr(path, recs):
    with open(path, 'wb') as fh:
        for name, s in recs:
            fh.write(b'>' + name + b'\n'); fh.write(s); fh.write(b'\n')
wr('/tmp/big1.fasta', [(b'big', A[rng.integers(0, 4, 50_000_000, dtype=np.uint8)].tobytes())])
wr('/tmp/short.fasta', [(b's%d' % i, A[rng.integers(0, 4, int(n), dtype=np.uint8)].tobytes()) for i, n in enumerate(rng.integers(300, 1400, 200_000))])
wr('/tmp/alln.fasta', [(b'n1', b'N' * 5000), (b'ok', A[rng.integers(0, 4, 6000, dtype=np.uint8)].tobytes()), (b'n2', b'n' * 1600),
                       (b'mixed', (b'ACGT' * 400 + b'N' * 3000 + b'ACGGT' * 300))])
PY
run() { local t0=$(date +%s%N); python -m jaeger_amd predict -i $1 -o /tmp/out_edge --model_path /tmp/model_brain --fsize 1500 --stride 1500 -f $2 2>&1 | grep -E "processed|rror|Traceback|wall time" | cut -c1-220; echo "  process wall: $(( ($(date +%s%N) - t0) / 1000000 )) ms"; }
echo "== one 50 Mbp contig"; run /tmp/big1.fasta ""; wc -l /tmp/out_edge/*/big1.tsv
echo "== 200 000 contigs of 300-1400 bp, --min-len 300"; run /tmp/short.fasta "--min-len 300"; wc -l /tmp/out_edge/*/short.tsv
echo "== same, --dust-host"; run /tmp/short.fasta "--min-len 300 --dust-host"
echo "== all-N and mixed contigs"; run /tmp/alln.fasta ""; cat /tmp/out_edge/*/alln.tsv | cut -f1-6
