"""TEST HARNESS: the product's device-decode cores (DEFLATE decoder, BAM framing / parsing) compiled for the host
as a wave of one lane (see decode_emu.cpp)."""
import ctypes as C
import os
import struct
import subprocess
import zlib

import numpy as np

from rnaseqc_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdecode_emu.so")
_ROOT = os.path.dirname(os.path.dirname(_HERE))


# "" = the product's configuration (one-pass commit of a round, long codes decoded inside the walk); the others flip the switches
VARIANTS = {"": (), "no_par_commit": ("-DINF_PAR_COMMIT_CFG=0",), "no_inwalk": ("-DINF_INWALK_CFG=0",),
            "all": ("-DINF_PAR_COMMIT_CFG=0", "-DINF_INWALK_CFG=0"),
            "vwalk1": ("-DINF_VWALK_CFG=1",), "vwalk2": ("-DINF_VWALK_CFG=2",), "vwalk3": ("-DINF_VWALK_CFG=3",),
            "vwalk3_no_inwalk": ("-DINF_VWALK_CFG=3", "-DINF_INWALK_CFG=0")}


def build(variant=""):
    """variant: a key of VARIANTS -- the decoder with the candidates of rsqc_inflate.h that are off in the product build
    (the one-pass commit of a round, the lane-parallel table build, long codes decoded inside the walk)."""
    csrc = os.path.join(_ROOT, "rnaseqc_amd", "csrc")
    srcs = [os.path.join(_HERE, "decode_emu.cpp")] + [os.path.join(csrc, h) for h in ("rsqc_inflate.h", "rsqc_bamrec.h", "rsqc_decode.h")] + \
           [os.path.join(_ROOT, "include", "rnaseqc_amd.h")]
    so = _SO.replace(".so", "_" + variant.replace("+", "_") + ".so") if variant else _SO
    if not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", *VARIANTS[variant], srcs[0], "-o", so])
    return so


def lib(variant=""):
    l = C.CDLL(build(variant))
    l.emu_inflate.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint32]
    l.emu_crc_wave64.restype = C.c_uint32
    l.emu_crc_wave64.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32]
    return l


def inflate(comp, n, crc, variant=""):
    """(status, bytes) of the wave-emulated DEFLATE decoder on one raw stream."""
    out = C.create_string_buffer(n + 64)
    rc = lib(variant).emu_inflate(comp, len(comp), out, n, crc & 0xFFFFFFFF)
    return rc, out.raw[:n]


class TagSpec(C.Structure):
    _fields_ = [("n_ref", C.c_int32), ("have_ch", C.c_uint8), ("ch0", C.c_uint8), ("ch1", C.c_uint8), ("n_filter", C.c_uint8),
                ("f0", C.c_uint8 * abi.MAX_FILTER_TAGS), ("f1", C.c_uint8 * abi.MAX_FILTER_TAGS)]


def tag_spec(n_ref, ch_tag="ch", filter_tags=()):
    t = TagSpec()
    t.n_ref = n_ref
    if ch_tag and len(ch_tag) == 2:
        t.have_ch, t.ch0, t.ch1 = 1, ord(ch_tag[0]), ord(ch_tag[1])
    for k, f in enumerate(filter_tags):
        t.f0[k], t.f1[k] = (ord(f[0]), ord(f[1])) if len(f) == 2 else (0, 0)
    t.n_filter = len(filter_tags)
    return t


def bgzf_blocks(path):
    """[(payload bytes, isize, crc)] of a BGZF file."""
    data = open(path, "rb").read()
    out, p = [], 0
    while p < len(data):
        xlen = struct.unpack_from("<H", data, p + 10)[0]
        o, bsize = 0, None
        while o < xlen:
            si1, si2, slen = struct.unpack_from("<BBH", data, p + 12 + o)
            if si1 == 66 and si2 == 67:
                bsize = struct.unpack_from("<H", data, p + 12 + o + 4)[0] + 1
            o += 4 + slen
        crc, isize = struct.unpack_from("<II", data, p + bsize - 8)
        out.append((data[p + 12 + xlen:p + bsize - 8], isize, crc))
        p += bsize
    return out


def inflate_bam(path, use_emu=True):
    """The inflated stream of a BAM file and the offset of its first record (after the header)."""
    parts = []
    for payload, isize, crc in bgzf_blocks(path):
        if use_emu:
            rc, b = inflate(payload, isize, crc)
            assert rc == 0, rc
        else:
            b = zlib.decompress(payload, -15)
        parts.append(b)
    s = b"".join(parts)
    assert s[:4] == b"BAM\1"
    l_text = struct.unpack_from("<i", s, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", s, p)[0]; p += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", s, p)[0]
        p += 8 + l_name
    return s, p, n_ref


class Decoded:
    pass


def decode_stream(stream, first, n_ref, window_bytes, ch_tag="ch", filter_tags=(), threads=7, perturb=0):
    """Runs the inflated stream through the emulated frame / chain / offsets / parse / lists steps in windows of
    `window_bytes`, carrying an incomplete last record over as the device decode does.  Returns the concatenated batch."""
    l = lib()
    tags = tag_spec(n_ref, ch_tag, filter_tags)
    carry = (C.c_int32 * 3)(0, 0, 0)
    cores, auxs, qh2s, cigs, seg_tid, seg_start, wide = [], [], [], [], [], [], []
    out = Decoded(); out.unsorted = False; out.bad_names = []; out.status = 0; out.windows = 0
    pos, tail, n_total, ops_total = first, b"", 0, 0
    while pos < len(stream) or tail:
        new = stream[pos:pos + window_bytes]; pos += len(new)
        buf = tail + new
        if not new and tail:
            raise RuntimeError("truncated BAM record")
        cap = len(buf) // 36 + 2
        core = np.zeros(cap, abi.REC_CORE); aux = np.zeros(cap, abi.REC_AUX); qh2 = np.zeros(cap, np.uint32); cig = np.zeros(len(buf) // 4 + 2, np.uint32)
        st = np.zeros(cap, np.int32); ss = np.zeros(cap + 1, np.uint64)
        wi = np.zeros(cap, np.uint64); wn = np.zeros(cap, np.int32); wl = np.zeros(cap, np.int32); wc = np.zeros(cap, np.uint32)
        summ = np.zeros(8 + 64, np.uint32)
        padded = buf + b"\0" * 64
        rc = l.emu_decode_window(padded, 0, len(buf), C.byref(tags), threads, carry, perturb, abi.ptr(core), abi.ptr(aux), abi.ptr(qh2), abi.ptr(cig),
                                 abi.ptr(st), abi.ptr(ss), abi.ptr(wi), abi.ptr(wn), abi.ptr(wl), abi.ptr(wc), abi.ptr(summ))
        assert rc == 0, "the listed repair and the sequential walk disagree (%d)" % rc
        n, ops, nseg, nwide, nbad, uns, consumed, status = [int(x) for x in summ[:8]]
        out.status |= status
        if status:
            break
        out.windows += 1
        core = core[:n].copy(); core["cigar_off"] += ops_total
        cores.append(core); auxs.append(aux[:n].copy()); qh2s.append(qh2[:n].copy()); cigs.append(cig[:ops].copy())
        for k in range(nseg):
            if seg_tid and k == 0 and seg_tid[-1] == int(st[0]):
                continue                                    # the window continues the previous one's contig
            seg_tid.append(int(st[k])); seg_start.append(n_total + int(ss[k]))
        for k in range(nwide):
            wide.append((n_total + int(wi[k]), int(wn[k]), int(wl[k]), int(wc[k])))
        for k in range(min(nbad, 64)):
            o = int(summ[8 + k]); ln = buf[o + 12]
            out.bad_names.append(buf[o + 36:o + 36 + ln].split(b"\0")[0].decode())
        out.unsorted |= bool(uns)
        n_total += n; ops_total += ops
        tail = buf[consumed:]
    out.core = np.concatenate(cores) if cores else np.zeros(0, abi.REC_CORE)
    out.aux = np.concatenate(auxs) if auxs else np.zeros(0, abi.REC_AUX)
    out.qhash2 = np.concatenate(qh2s) if qh2s else np.zeros(0, np.uint32)
    out.cigar = np.concatenate(cigs) if cigs else np.zeros(0, np.uint32)
    out.seg_tid = np.array(seg_tid, np.int32); out.seg_start = np.array(seg_start + [n_total], np.uint64)
    out.wide = wide; out.n = n_total
    return out


from rnaseqc_amd.bamio import feed_chunks, _feed_lib, BGZF_BLOCK as BLOCK  # noqa: E402,F401  (the feeder's Python binding)
