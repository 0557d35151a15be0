# where does the host time of a 500-bp end-to-end run go?  cProfile of the CLI on the 407 Mbp workload
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/e2e
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
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
sys.path.insert(0, 'tests')
from conftest import make_model_dir
make_model_dir(Path('/tmp/model_b500'), name="baseline500", model_name="jaeger_500bp_baseline")
PY
python -m jaeger_amd predict -i /tmp/synth10k.fasta -o /tmp/out_w --model_path /tmp/model_b500 --fsize 500 --stride 500 -f > /dev/null 2>&1
python -X importtime -c "import jaeger_amd.cli" 2> /tmp/imp.txt; sort -t'|' -k2 -n /tmp/imp.txt | tail -8
python -m cProfile -o /tmp/prof.out -m jaeger_amd predict -i /tmp/synth10k.fasta -o /tmp/out_p --model_path /tmp/model_b500 --fsize 500 --stride 500 -f --no-pipeline 2>&1 | grep -E "INFO" | cut -c1-200 | tail -20
python - <<'PY'
import pstats
p = pstats.Stats('/tmp/prof.out')
p.sort_stats('cumulative').print_stats(45)
PY
