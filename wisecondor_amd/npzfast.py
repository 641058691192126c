"""np.savez_compressed with the deflate spread over threads.

Once the GPU side of `newref` takes milliseconds, the sub-command's wall time is zlib: the
reference writes its prep, part and reference files with np.savez_compressed (wisecondor.py:97-108,
128-132, 160-170) -- one thread, level 6, ~30 MB/s: 1.7 of the 1.8 s of `newref` at 100 samples x
250 kb, half a minute at 600 x 50 kb.  This writer produces the same container (a zip of .npy
members, read back by np.load exactly like numpy's own files) but cuts every large member into
chunks that are deflated independently in a thread pool (zlib releases the GIL) and concatenated
into ONE raw deflate stream: every chunk but the last ends with a sync flush -- byte aligned, no
final-block flag -- the way pigz does it.  Nothing numeric happens here.
"""
import concurrent.futures
import io
import struct
import zlib

import numpy as np

CHUNK = 4 << 20          # bytes of a member deflated by one task


def _npy_bytes(value):
    """The .npy image np.savez would store for `value` (object arrays pickled, as numpy does)."""
    buf = io.BytesIO()
    np.lib.format.write_array(buf, np.asanyarray(value), allow_pickle=True)
    return buf.getbuffer()


def _deflate_chunk(view, level, last):
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    out = c.compress(view)
    return out + c.flush(zlib.Z_FINISH if last else zlib.Z_SYNC_FLUSH)


def savez(path, level=6, threads=8, **arrays):
    """Write `arrays` as <path> (an .npz; '.npz' is appended like numpy does when missing).
    level 0 stores the members; otherwise they are deflated at `level` in `threads` threads."""
    path = str(path)
    if not path.endswith('.npz'):
        path += '.npz'
    members = [(name + '.npy', _npy_bytes(value)) for name, value in arrays.items()]
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, int(threads))) as pool:
        jobs = []
        for name, raw in members:
            if level <= 0 or len(raw) < 256:
                jobs.append((name, raw, None))
                continue
            view = memoryview(raw)
            cuts = list(range(0, len(view), CHUNK))
            parts = [pool.submit(_deflate_chunk, view[a:a + CHUNK], level, a == cuts[-1]) for a in cuts]
            jobs.append((name, raw, parts))
        with open(path, 'wb') as f:
            central = []
            for name, raw, parts in jobs:
                crc = zlib.crc32(raw) & 0xFFFFFFFF
                body = [raw] if parts is None else [p.result() for p in parts]
                csize = sum(len(b) for b in body)
                usize = len(raw)
                method = 0 if parts is None else 8
                offset = f.tell()
                fname = name.encode()
                big = csize >= 0xFFFFFFFF or usize >= 0xFFFFFFFF or offset >= 0xFFFFFFFF
                extra = struct.pack('<HHQQ', 1, 16, usize, csize) if big else b''
                f.write(struct.pack('<IHHHHHIIIHH', 0x04034b50, 45 if big else 20, 0, method, 0, 0x21, crc,
                                    0xFFFFFFFF if big else csize, 0xFFFFFFFF if big else usize, len(fname), len(extra)))
                f.write(fname)
                f.write(extra)
                for b in body:
                    f.write(b)
                central.append((fname, method, crc, csize, usize, offset, big))
            cd_start = f.tell()
            for fname, method, crc, csize, usize, offset, big in central:
                extra = struct.pack('<HHQQQ', 1, 24, usize, csize, offset) if big else b''
                f.write(struct.pack('<IHHHHHHIIIHHHHHII', 0x02014b50, 45, 45 if big else 20, 0, method, 0, 0x21, crc,
                                    0xFFFFFFFF if big else csize, 0xFFFFFFFF if big else usize, len(fname), len(extra),
                                    0, 0, 0, 0x01800000, 0xFFFFFFFF if big else offset))
                f.write(fname)
                f.write(extra)
            cd_size = f.tell() - cd_start
            if cd_start >= 0xFFFFFFFF or len(central) >= 0xFFFF:
                z64 = f.tell()
                f.write(struct.pack('<IQHHIIQQQQ', 0x06064b50, 44, 45, 45, 0, 0, len(central), len(central), cd_size,
                                    cd_start))
                f.write(struct.pack('<IIQI', 0x07064b50, 0, z64, 1))
                f.write(struct.pack('<IHHHHIIH', 0x06054b50, 0, 0, 0xFFFF, 0xFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0))
            else:
                f.write(struct.pack('<IHHHHIIH', 0x06054b50, 0, 0, len(central), len(central), cd_size, cd_start, 0))
    return path
