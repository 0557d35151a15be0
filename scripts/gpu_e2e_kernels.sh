# kernel statistics of an end-to-end CLI run (500-bp family, 407.6 Mbp FASTA, DUST on the GPU): rocprofv3 --kernel-trace --stats
cd $GRAFT_REPO_ROOT
O=gpurun_out/e2e
mkdir -p $O
export TMPDIR=/tmp
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from bench import synth_contigs
rng = np.random.Generator(np.random.PCG64(20260923))
lengths, bases = synth_contigs(rng, 10000)
with open('/tmp/synth10k.fasta', 'wb') as fh:
    off = 0
    for i, l in enumerate(lengths):
        fh.write(b'>contig_%d len=%d\n' % (i, l))
        s = bases[off:off + l].tobytes(); off += l
        fh.write(b'\n'.join(s[j:j + 80] for j in range(0, l, 80)) + b'\n')
from pathlib import Path
from conftest import make_model_dir
make_model_dir(Path('/tmp/model_b500'), name="baseline500", model_name="jaeger_500bp_baseline")
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o e2e -- python3 -m jaeger_amd predict -i /tmp/synth10k.fasta -o /tmp/out_k --model_path /tmp/model_b500 --fsize 500 --stride 500 -f 2>&1 | grep -E "wall time|DUST" | cut -c1-260
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $O/e2e_baseline500_kernel_stats.csv
head -9 $O/e2e_baseline500_kernel_stats.csv | cut -c1-170
rm -rf $O/prof
