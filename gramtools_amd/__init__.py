"""gramtools_amd — MI355X-native quasimap engine for gramtools (hot path only).

The package holds the HIP kernels + C-ABI (``csrc/``, built in-tree into ``lib/libgmx.so``), the
``gram`` drop-in executable (``bin/gram``) and this thin Python mirror of the reference's quasimap
interface. See DESIGN.md for the path, the boundary and the data layout.
"""
from .quasimap import (  # noqa: F401
    Index,
    Quasimapper,
    PackedReads,
    Ingest,
    bgzf_members,
    PinnedArray,
    pack_reads, pack_reads_2bit,
    QuasimapperGroup,
    Coverage,
    QuasimapReadsStats,
    quasimap_reads,
    Genotyped,
    genotyping_model,
    genotyping_model_debug,
    master_seeds,
    encode_dna_bases,
    dump_allele_sum,
    dump_allele_base,
    allele_base_json,
    hash_allele_groups,
    group_id_counts,
    group_id_alleles,
    grouped_json,
    dump_grouped_allele_counts,
    RNG_LEMIRE,
    GMX_INGEST_BAD_RECORD, GMX_INGEST_BAD_MEMBER, GMX_INGEST_BAD_CRC, GMX_INGEST_TOO_MANY_LINES,
    RNG_DIVISION,
)
from ._lib import GmxError  # noqa: F401
