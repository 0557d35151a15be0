# end-to-end CLI run on a synthetic 10 000-contig FASTA (BASELINE configs[1] shape), stage timings from the log
python - <<'PY'
import numpy as np, sys, time
sys.path.insert(0, '.')
from bench import synth_contigs
rng = np.random.Generator(np.random.PCG64(20260923))
lengths, bases = synth_contigs(rng, 10000)
t = time.time()
with open('/tmp/synth10k.fasta', 'wb') as fh:
    off = 0
    for i, l in enumerate(lengths):
        fh.write(b'>contig_%d len=%d\n' % (i, l))
        s = bases[off:off + l].tobytes(); off += l
        fh.write(b'\n'.join(s[j:j + 80] for j in range(0, l, 80)) + b'\n')
print('wrote', off / 1e6, 'Mbp in', time.time() - t)
import yaml, shutil
from pathlib import Path
sys.path.insert(0, 'tests')
from conftest import make_model_dir
make_model_dir(Path('/tmp/model_brain'))
PY
time python -m jaeger_amd predict -i /tmp/synth10k.fasta -o /tmp/out_e2e --model_path /tmp/model_brain --fsize 1500 --stride 1500 --no-dustmask -f 2>&1 | grep -E "wall time|processed|error|GPU worker"
time python -m jaeger_amd predict -i /tmp/synth10k.fasta -o /tmp/out_e2e --model_path /tmp/model_brain --fsize 1500 --stride 1500 -f 2>&1 | grep -E "wall time|processed|error|DUST|GPU worker" 
cp /tmp/out_e2e/*/synth10k.tsv /tmp/piped.tsv
time python -m jaeger_amd predict -i /tmp/synth10k.fasta -o /tmp/out_e2e --model_path /tmp/model_brain --fsize 1500 --stride 1500 -f --no-pipeline 2>&1 | grep -E "wall time|processed|error|DUST|GPU worker"
cmp /tmp/piped.tsv /tmp/out_e2e/*/synth10k.tsv && echo "pipelined and sequential TSVs identical"
wc -l /tmp/out_e2e/*/synth10k.tsv
grep -h "terminal repeats\|DUST" /tmp/out_e2e/*/*_jaeger.log | tail -3
