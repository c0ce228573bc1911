"""Device-side ingestion of a BGZF FASTQ (gmx_ingest_*): rate of the whole chain (upload, inflate + CRC, record scan, packing)
over a file handed over in chunks of members, two slots alternating — alone and with the reads mapped (configs[1]).
Usage: python tools/ingest_bench.py [n_reads] [quality model: binned|wide|const] [members per chunk]"""
import os, struct, sys, time, zlib
from concurrent.futures import ProcessPoolExecutor
import numpy as np
sys.path.insert(0, ".")

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
QUAL = sys.argv[2] if len(sys.argv) > 2 else "binned"
STEP = int(sys.argv[3]) if len(sys.argv) > 3 else 8000
L = 150


def make_text(seed, first, n, reads=None):
    """n Illumina-style records: instrument:run:flowcell:lane:tile:x:y header, `reads` (uint8 1..4) or random bases, qualities by model."""
    rng = np.random.default_rng(seed)
    if reads is None:
        reads = rng.integers(1, 5, size=(n, L), dtype=np.uint8)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[reads - 1]
    if QUAL == "const":
        q = np.full((n, L), ord("I"), dtype=np.uint8)
    elif QUAL == "binned":  # NovaSeq-like: four levels, mostly the best, in runs
        lvl = np.frombuffer(b"F:,#", dtype=np.uint8)
        pick = rng.choice(4, size=(n, L // 5), p=[0.9, 0.06, 0.03, 0.01])
        q = lvl[np.repeat(pick, 5, axis=1)]
    else:  # forty levels, drifting down along the read
        base = 40 - (np.arange(L) * 12 // L)[None, :] - rng.integers(0, 8, size=(n, L))
        q = (33 + np.clip(base, 2, 40)).astype(np.uint8)
    xs, ys = rng.integers(1000, 30000, n), rng.integers(1000, 30000, n)
    out = []
    for i in range(n):
        out.append(b"@A00123:45:HXXXXXXXX:1:%d:%d:%d 1:N:0:ACGTACGT\n" % (1101 + (first + i) // 40000, xs[i], ys[i]))
        out.append(bases[i].tobytes())
        out.append(b"\n+\n")
        out.append(q[i].tobytes())
        out.append(b"\n")
    return b"".join(out)


def bgzf_piece(args):
    seed, first, n = args
    text = make_text(seed, first, n)
    out = bytearray()
    for i in range(0, len(text), 65280):
        piece = text[i:i + 65280]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = c.compress(piece) + c.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(comp) + 8 - 1)
        out += comp + struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece))
    return bytes(out), len(text)


if __name__ == "__main__":
    t0 = time.time()
    per = 50000
    jobs = [(1000 + i, i * per, min(per, N - i * per)) for i in range((N + per - 1) // per)]
    with ProcessPoolExecutor(max_workers=min(32, os.cpu_count() or 8)) as ex:
        parts = list(ex.map(bgzf_piece, jobs))
    data = b"".join(p[0] for p in parts)
    text_bytes = sum(p[1] for p in parts)
    print(f"{N} reads, qualities '{QUAL}': {text_bytes / 1e6:.0f} MB of text, {len(data) / 1e6:.0f} MB of BGZF ({len(data) / N:.1f} B/read) in {time.time() - t0:.0f} s", flush=True)
    import ctypes as C
    from gramtools_amd import Ingest, bgzf_members, PinnedArray
    mem = bgzf_members(data)
    print(f"{len(mem)} members, {STEP} per chunk", flush=True)
    pin = PinnedArray(len(data) + 64, np.uint8)
    pin.array[:len(data)] = np.frombuffer(data, dtype=np.uint8)
    ing = Ingest(max_text_bytes=STEP * 65536 + (1 << 20))
    chunks = [mem[i:i + STEP] for i in range(0, len(mem), STEP)]
    # (the member tables as the C ABI takes them, built ahead: `gram` walks the file's member table on a thread of its own)
    arrays = [Ingest.member_array([(o - ch[0][0], s, i, c) for o, s, i, c in ch]) for ch in chunks]

    def run(mapper=None, seeds=None):
        ing.reset()
        total = 0
        pending = []

        def submit(ci):
            ch = chunks[ci]
            lo, hi = ch[0][0], ch[-1][0] + ch[-1][1]
            ing.submit_bgzf(ci % 3, pin.array[lo:hi], arrays[ci], ci == len(chunks) - 1)
        t = time.perf_counter()
        submit(0)
        if len(chunks) > 1:
            submit(1)
        for ci in range(len(chunks)):
            if ci + 2 < len(chunks):  # (three slots: chunk ci + 2 takes chunk ci - 1's, waited for and released)
                submit(ci + 2)
            res = ing.wait(ci % 3)
            assert res.status == 0, (res.status, res.bad_member)
            if mapper is not None:
                mapper.map_ingested(res, seeds)
                ing.release_after(ci % 3, engine=qm)
            total += int(res.n_reads)
        if mapper is not None:
            mapper.sync()
        return total, time.perf_counter() - t
    for rep in range(3):
        n, dt = run()
        print(f"ingest alone: {n} reads in {dt * 1e3:.1f} ms = {n / dt / 1e6:.1f} M reads/s ({text_bytes / dt / 1e9:.2f} GB/s of text, {len(data) / dt / 1e9:.2f} GB/s compressed)", flush=True)
    if os.environ.get("INGEST_MAP", "1") == "1":
        from gramtools_amd import Index, Quasimapper
        from gramtools_amd.synth import random_ref, snp_prg
        ref = random_ref(4411532, 1)
        prg, pos, alts, n_alts = snp_prg(ref, 60000, 2)
        qm = Quasimapper(Index(prg, 10))
        seeds = PinnedArray(STEP * 400, np.uint32)
        seeds.array[:] = 12345
        for rep in range(2):
            qm.reset()
            n, dt = run(qm, seeds)
            print(f"ingest + quasimap (random reads: none maps): {n} reads in {dt * 1e3:.1f} ms = {n / dt / 1e6:.1f} M reads/s", flush=True)
