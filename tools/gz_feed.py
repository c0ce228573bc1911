"""Gzip feeds of `gram` (gmx_gzsource.h): a 4 M-read FASTQ as plain gzip (all threads / zlib alone) and as BGZF —
decompression alone (`gram _gz_info`) and the parser behind it (`gram _parse_bench`). Usage: python tools/gz_feed.py"""
import gzip, os, struct, subprocess, sys, time, zlib
import numpy as np
sys.path.insert(0, ".")
from bench import write_fastq
reads = np.random.default_rng(1).integers(1, 5, size=(4000000, 150), dtype=np.uint8)
write_fastq("/tmp/r4m.fq", [reads])
raw = open("/tmp/r4m.fq", "rb").read()
t = time.time(); subprocess.check_call("gzip -k -6 -f /tmp/r4m.fq", shell=True); print(f"gzip -6: {os.path.getsize('/tmp/r4m.fq.gz') / 1e6:.0f} MB of {len(raw) / 1e6:.0f} MB in {time.time() - t:.0f} s")
out = bytearray()
for i in range(0, len(raw), 65280):
    piece = raw[i:i + 65280]
    c = zlib.compressobj(6, zlib.DEFLATED, -15); comp = c.compress(piece) + c.flush()
    out += b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(comp) + 8 - 1)
    out += comp + struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece))
open("/tmp/r4m.bgzf.fq.gz", "wb").write(bytes(out))
gram = "gramtools_amd/bin/gram"
for name, path, env in (("plain gzip, all threads", "/tmp/r4m.fq.gz", {}), ("plain gzip, zlib alone (GMX_PARGZ=0)", "/tmp/r4m.fq.gz", {"GMX_PARGZ": "0"}),
                        ("BGZF", "/tmp/r4m.bgzf.fq.gz", {})):
    for threads in (16, 64):
        best = None
        for rep in range(3):
            o = subprocess.run([gram, "_gz_info", path, str(threads)], stdout=subprocess.PIPE, text=True, env=dict(os.environ, **env)).stdout.strip()
            kv = dict(x.split("=") for x in o.split())
            if best is None or float(kv["seconds"]) < float(best["seconds"]): best = kv
        print(f"{name}, {threads} threads: decompress + CRC {float(best['seconds']):.3f} s = {float(best['MBps']) / 1e3:.2f} GB/s of text = "
              f"{4e6 / float(best['seconds']) / 1e6:.1f} M reads/s (pieces {best['pieces']}, bgzf members {best['bgzf_members']}, zlib bytes {best['stream_bytes']})")
    for rep in range(2):
        o = subprocess.run([gram, "_parse_bench", path, "64", "2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, **env)).stdout
        print("   parse bench:", " | ".join(o.strip().splitlines()[-2:]))

o = subprocess.run([gram, "_gz_info", "/tmp/r4m.fq.gz", "16"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, GMX_PARGZ_TRACE="1", GMX_GZ_INFO_NO_CRC="1")).stdout
print("phases of the rounds (16 threads):")
print("\n".join(o.strip().splitlines()[:6]))
for f in ("/tmp/r4m.fq", "/tmp/r4m.fq.gz", "/tmp/r4m.bgzf.fq.gz"):
    os.remove(f)
