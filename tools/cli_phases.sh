#!/bin/bash
# Where one whole `gram genotype` call on the bench workload spends its wall time (GMX_PHASE_TRACE=1), after a warm-up call.
# Usage: tools/cli_phases.sh [N_READS=4000000] [THREADS=64]
N=${1:-4000000}; T=${2:-64}
D=$(mktemp -d /tmp/cliphase.XXXX)
python - "$N" "$D" <<'PY'
import sys, os
import numpy as np
sys.path.insert(0, ".")
from bench import write_fastq, GENOME, N_SITES, KMER
from gramtools_amd.synth import random_ref, snp_prg, simulate_snp_reads_fast
n, d = int(sys.argv[1]), sys.argv[2]
ref = random_ref(GENOME, 1)
prg, pos, alts, n_alts = snp_prg(ref, N_SITES, 2)
np.asarray(prg, dtype="<u4").tofile(os.path.join(d, "prg"))
write_fastq(os.path.join(d, "r.fq"), [simulate_snp_reads_fast(ref, pos, alts, n_alts, n, 150, 1000)])
PY
G=gramtools_amd/bin/gram
$G build --gram_dir $D --kmer_size 10 --max_threads $T > /dev/null
for rep in 0 1 2; do
  s=$(date +%s.%N)
  GMX_PHASE_TRACE=1 $G genotype --gram_dir $D --reads $D/r.fq --sample_id s --ploidy haploid --kmer_size 10 --genotype_dir $D/run$rep --max_threads $T --seed 42 2> $D/ph$rep.txt > $D/out$rep.txt
  e=$(date +%s.%N)
  echo "call $rep: whole call $(python -c "print(f'{($e - $s) * 1e3:.0f}')") ms wall (process start to exit)"
done
cat $D/ph2.txt
grep -E "Load data|Quasimap|Genotyping" $D/out2.txt
rm -rf $D
