#!/bin/bash
# `gram genotype --devices 0,0,0,0` on a plain FASTQ, phases and feed trace: where the wall time of the dealt text route goes.
set -u
export TMPDIR=/tmp
D=$(mktemp -d /tmp/gmx_dealt.XXXX)
python - "$D" <<'PY'
import sys, os, numpy as np
sys.path.insert(0, ".")
from bench import write_fastq, GENOME, N_SITES
from gramtools_amd.synth import random_ref, snp_prg, simulate_snp_reads_fast
d = sys.argv[1]
ref = random_ref(GENOME, 1); prg, pos, alts, n_alts = snp_prg(ref, N_SITES, 2)
np.asarray(prg, dtype="<u4").tofile(os.path.join(d, "prg"))
write_fastq(os.path.join(d, "r.fq"), [simulate_snp_reads_fast(ref, pos, alts, n_alts, 1_000_000, 150, 1000 + i) for i in range(4)])
PY
G=gramtools_amd/bin/gram
ls -la $D; $G build --gram_dir $D --kmer_size 10 --max_threads 16 | tail -2
for devs in 0 0,0,0,0; do for th in 8 64; do
  echo "== --devices $devs --max_threads $th"
  T0=$(date +%s.%N)
  GMX_PHASE_TRACE=1 $G genotype --gram_dir $D --reads $D/r.fq --sample_id s --ploidy haploid --kmer_size 10 --genotype_dir $D/out --max_threads $th --seed 1 --devices $devs > $D/log.txt 2>&1
  echo "rc=$? wall $(python3 -c "import time,sys; print(round(time.time()-float(sys.argv[1]),2))" $T0) s"
  grep -E "phase|Quasimap \(parse|warning|rror" $D/log.txt | cut -c1-160
done; done
rm -rf $D
