"""Minimal wav I/O with soundfile's semantics (the reference's drivers use `sf.read` / `sf.write`,
scripts/evaluate_M1.py:114,165-166): RIFF/WAVE PCM-16 mono, and -- for the raw WSJ0 files the reference's
data-set scripts read (scripts/create_test_set.py via libsndfile) -- NIST SPHERE headers with uncompressed
16-bit PCM.  read -> float64 = int16 / 32768;
write(float) -> PCM-16 = round(x * 32767) clipped (libsndfile's normalised float conversion).
soundfile itself is not installable here, so the writer's rounding rule is "parity unpinned"."""
import struct

import numpy as np


def read(path):
    """Returns (float64 samples in [-1,1), sample rate)."""
    b = open(path, "rb").read()
    if b[:7] == b"NIST_1A":
        return _read_sphere(path, b)
    if b[:4] != b"RIFF" or b[8:12] != b"WAVE":
        raise ValueError("%s: not a RIFF/WAVE file" % path)
    pos, fs = 12, None
    while pos + 8 <= len(b):
        cid, sz = b[pos:pos + 4], struct.unpack("<I", b[pos + 4:pos + 8])[0]
        if cid == b"fmt ":
            fmt, ch, fs = struct.unpack("<HHI", b[pos + 8:pos + 16])
            bits = struct.unpack("<H", b[pos + 22:pos + 24])[0]
            if fmt != 1 or ch != 1 or bits != 16:
                raise NotImplementedError("%s: only PCM-16 mono is supported (fmt=%d ch=%d bits=%d)" % (path, fmt, ch, bits))
        elif cid == b"data":
            if fs is None:
                raise ValueError("%s: data chunk before fmt chunk" % path)
            pcm = np.frombuffer(b[pos + 8:pos + 8 + sz], dtype="<i2")
            return pcm.astype(np.float64) / 32768.0, fs
        pos += 8 + sz + (sz & 1)
    raise ValueError("%s: no data chunk" % path)


def _read_sphere(path, b):
    """NIST SPHERE: 'NIST_1A\n<header bytes>\n' then `key -type value` lines up to end_head; samples follow the header."""
    hsize = int(b[8:16].decode("ascii").strip())
    fields = {}
    for line in b[16:hsize].decode("ascii", "replace").split("\n"):
        parts = line.split(None, 2)
        if not parts or parts[0] == "end_head":
            break
        if len(parts) == 3:
            fields[parts[0]] = parts[2].strip()
    coding = fields.get("sample_coding", "pcm")
    if coding != "pcm" or int(fields.get("sample_n_bytes", 2)) != 2 or int(fields.get("channel_count", 1)) != 1:
        raise NotImplementedError("%s: only uncompressed 16-bit mono SPHERE files are supported (coding=%s)" % (path, coding))
    n = int(fields["sample_count"])
    order = "<i2" if fields.get("sample_byte_format", "01") == "01" else ">i2"
    pcm = np.frombuffer(b[hsize:hsize + 2 * n], dtype=order)
    if len(pcm) != n:
        raise ValueError("%s: %d samples announced, %d present" % (path, n, len(pcm)))
    return pcm.astype(np.float64) / 32768.0, int(fields["sample_rate"])


def write(path, x, fs):
    x = np.asarray(x, dtype=np.float64)
    pcm = np.clip(np.rint(x * 32767.0), -32768, 32767).astype("<i2")
    data = pcm.tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, int(fs), int(fs) * 2, 2, 16)
    with open(path, "wb") as f:
        f.write(hdr + b"data" + struct.pack("<I", len(data)) + data)
