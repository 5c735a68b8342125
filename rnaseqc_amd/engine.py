"""Host-side driver of the HIP hot path through the C ABI (include/rnaseqc_amd.h).

This is plumbing: it loads rnaseqc_amd/lib/librnaseqc_amd.so (built by
__graft_entry__.build() / rnaseqc_amd/csrc/Makefile) and forwards calls.  There is no
fallback of any kind: a missing library or a missing GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RSQC_LIB") or os.path.join(_HERE, "lib", "librnaseqc_amd.so")   # RSQC_LIB: a diagnostic build (make prof)
_lib = None
_hip = None


class EngineError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("rnaseqc_amd error %d: %s" % (code, msg))
        self.code = code


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError(abi.ERR_NO_DEVICE, "HIP library %s is missing; run __graft_entry__.build()" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        lib.rsqc_create.argtypes = [C.POINTER(abi.Params), C.POINTER(vp)]
        lib.rsqc_destroy.argtypes = [vp]; lib.rsqc_destroy.restype = None
        lib.rsqc_set_annotation.argtypes = [vp, C.POINTER(abi.AnnotationStruct), vp]
        lib.rsqc_set_bed.argtypes = [vp, C.POINTER(abi.BedStruct)]
        lib.rsqc_set_reference.argtypes = [vp, C.POINTER(abi.ReferenceStruct)]
        lib.rsqc_submit.argtypes = [vp, C.POINTER(abi.BatchStruct)]
        lib.rsqc_wait.argtypes = [vp]
        lib.rsqc_upload.argtypes = [vp, C.POINTER(abi.BatchStruct), C.POINTER(C.c_int)]
        lib.rsqc_submit_resident.argtypes = [vp, C.c_int]
        lib.rsqc_release.argtypes = [vp, C.c_int]
        lib.rsqc_finalize.argtypes = [vp, C.POINTER(abi.ResultsStruct)]
        lib.rsqc_reset.argtypes = [vp]
        lib.rsqc_get_timing.argtypes = [vp, C.POINTER(abi.TimingStruct)]
        lib.rsqc_reset_timing.argtypes = [vp]
        lib.rsqc_device_accumulators.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint64), C.POINTER(vp), C.POINTER(C.c_uint64)]
        lib.rsqc_device_vectors.argtypes = [vp, C.POINTER(abi.DeviceRange * 3)]
        lib.rsqc_shard_summary.argtypes = [vp, C.POINTER(abi.ShardInfoStruct)]
        lib.rsqc_reduce_peer.argtypes = [vp, vp]
        lib.rsqc_reduce_group.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_int)]
        lib.rsqc_group_create.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(vp)]
        lib.rsqc_group_reduce.argtypes = [vp, C.POINTER(C.c_int)]
        lib.rsqc_group_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_char_p)]
        lib.rsqc_group_destroy.argtypes = [vp]; lib.rsqc_group_destroy.restype = None
        lib.rsqc_refresh_results.argtypes = [vp, C.POINTER(abi.ResultsStruct)]
        lib.rsqc_finalize_device.argtypes = [vp]
        lib.rsqc_host_alloc.argtypes = [C.c_size_t]; lib.rsqc_host_alloc.restype = vp
        lib.rsqc_host_free.argtypes = [vp]; lib.rsqc_host_free.restype = None
        lib.rsqc_strerror.argtypes = [C.c_int]; lib.rsqc_strerror.restype = C.c_char_p
        lib.rsqc_last_error.argtypes = [vp]; lib.rsqc_last_error.restype = C.c_char_p
        lib.rsqc_counter_name.argtypes = [C.c_int]; lib.rsqc_counter_name.restype = C.c_char_p
        lib.rsqc_version.restype = C.c_char_p
        lib.rsqc_qname_hash.argtypes = [C.c_char_p, C.c_size_t]; lib.rsqc_qname_hash.restype = C.c_uint64
        lib.rsqc_qname_hash2.argtypes = [C.c_char_p, C.c_size_t]; lib.rsqc_qname_hash2.restype = C.c_uint32
        lib.rsqc_decode_begin.argtypes = [vp, C.POINTER(abi.DecodeParams)]
        lib.rsqc_decode_submit.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(abi.DecodeWindow)]
        lib.rsqc_decode_end.argtypes = [vp, C.POINTER(abi.DecodeInfo)]
        _lib = lib
    return _lib


EXPORTED_SYMBOLS = [
    "rsqc_create", "rsqc_destroy", "rsqc_set_annotation", "rsqc_set_bed", "rsqc_set_reference", "rsqc_submit", "rsqc_wait",
    "rsqc_upload", "rsqc_submit_resident", "rsqc_release", "rsqc_finalize", "rsqc_reset", "rsqc_get_timing",
    "rsqc_reset_timing", "rsqc_device_accumulators", "rsqc_device_vectors", "rsqc_shard_summary", "rsqc_reduce_peer", "rsqc_reduce_group",
    "rsqc_group_create", "rsqc_group_reduce", "rsqc_group_info", "rsqc_group_destroy", "rsqc_refresh_results", "rsqc_finalize_device", "rsqc_host_alloc", "rsqc_host_free", "rsqc_strerror",
    "rsqc_last_error", "rsqc_counter_name", "rsqc_version", "rsqc_qname_hash", "rsqc_qname_hash2",
    "rsqc_decode_begin", "rsqc_decode_submit", "rsqc_decode_end",
]


class DeviceArray:
    """A raw HIP device pointer exposed through __cuda_array_interface__ so that
    torch.as_tensor(..., device='cuda') wraps it zero-copy (used for the RCCL reduction)."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class Engine:
    def __init__(self, params: abi.Params):
        self._l = load_library()
        self._h = C.c_void_p()
        self._keep = []
        self._pinned = []
        rc = self._l.rsqc_create(C.byref(params), C.byref(self._h))
        if rc:
            raise EngineError(rc, self._l.rsqc_strerror(rc).decode())

    def _check(self, rc):
        if rc:
            raise EngineError(rc, "%s (%s)" % (self._l.rsqc_strerror(rc).decode(),
                                               self._l.rsqc_last_error(self._h).decode()))

    def last_error(self) -> str:
        """rsqc_last_error: the message of the last failure -- or the WARNING an accepted annotation left (an exon outside its gene's row)."""
        return self._l.rsqc_last_error(self._h).decode()

    def set_annotation(self, ann, owned=None):
        s = ann.to_struct()
        o = None if owned is None else np.ascontiguousarray(owned, dtype=np.uint8)
        self._keep += [ann, s, o]
        self._check(self._l.rsqc_set_annotation(self._h, C.byref(s), abi.ptr(o)))
        # RSQC_OK may come with a warning (an exon row outside the row of its gene: DESIGN.md 5): returned, kept and logged here,
        # because rsqc_last_error is overwritten by any later failure; the results carry the count (exons_outside_gene_row)
        self.annotation_warning = self.last_error() or None
        if self.annotation_warning:
            import warnings
            warnings.warn("rsqc_set_annotation: " + self.annotation_warning, RuntimeWarning, stacklevel=2)
        return self.annotation_warning

    def set_bed(self, bed):
        s = bed.to_struct()
        self._keep += [bed, s]
        self._check(self._l.rsqc_set_bed(self._h, C.byref(s)))

    def set_reference(self, ref):
        s = ref.to_struct()
        self._check(self._l.rsqc_set_reference(self._h, C.byref(s)))     # (copied: nothing of `ref` is kept)

    def submit(self, batch):
        s = batch.to_struct()
        self._keep += [batch, s]
        self._check(self._l.rsqc_submit(self._h, C.byref(s)))

    def submit_struct(self, s):
        """rsqc_submit of an already packed abi.BatchStruct (the caller keeps its arrays alive until wait())."""
        self._check(self._l.rsqc_submit(self._h, C.byref(s)))

    def pinned_copy(self, a: np.ndarray) -> np.ndarray:
        """A copy of `a` in page-locked host memory (rsqc_host_alloc); freed with the engine."""
        a = np.ascontiguousarray(a)
        nbytes = max(a.nbytes, 16)
        p = self._l.rsqc_host_alloc(nbytes)
        if not p:
            raise EngineError(abi.ERR_HIP, "rsqc_host_alloc failed")
        self._pinned.append(p)
        out = np.frombuffer((C.c_char * nbytes).from_address(p), dtype=a.dtype, count=a.size).reshape(a.shape)
        out[...] = a
        return out

    def wait(self):
        self._check(self._l.rsqc_wait(self._h))

    # ---- device-side BAM decode (rsqc_decode_*) -----------------------------------------------------------------
    def decode_begin(self, n_ref, ch_tag="ch", filter_tags=(), file_index_base=0, pipelined=False, reserve=0):
        p = abi.DecodeParams()
        p.pipelined = 1 if pipelined else 0
        p.reserve_inflated_bytes = reserve
        p.n_ref = n_ref
        if ch_tag and len(ch_tag) == 2:
            p.has_chimeric_tag = 1; p.chimeric_tag = ch_tag.encode()
        for k, f in enumerate(filter_tags):
            if len(f) == 2:
                p.filter_tag[k].value = f.encode()
        p.file_index_base = file_index_base
        self._check(self._l.rsqc_decode_begin(self._h, C.byref(p)))

    def decode_submit(self, compressed, blocks, skip=0, limit=0):
        """compressed: bytes-like or (address, nbytes); blocks: numpy array of abi BGZF block records (or (address, n)).
        Returns (records, [RefID of every run of records])."""
        if isinstance(compressed, tuple):
            caddr, cbytes = compressed
        else:
            buf = np.frombuffer(compressed, np.uint8)
            self._keep_decode = buf
            caddr, cbytes = buf.ctypes.data, buf.size
        if isinstance(blocks, tuple):
            baddr, nb = blocks
        else:
            blocks = np.ascontiguousarray(blocks)
            baddr, nb = blocks.ctypes.data, len(blocks)
        w = abi.DecodeWindow()
        self._check(self._l.rsqc_decode_submit(self._h, caddr, cbytes, baddr, nb, skip, limit, C.byref(w)))
        self._last_window = w
        runs = list(np.ctypeslib.as_array(C.cast(w.run_tid, C.POINTER(C.c_int32)), (w.n_runs,))) if w.n_runs else []
        return int(w.n_records), [int(t) for t in runs]

    def decode_end(self):
        """(records, unsorted, number of records with an unrecognised RefID, first names of those); .decode_last = (records,
        runs) of the call that rsqc_decode_end completed (pipelined streams)"""
        info = abi.DecodeInfo()
        rc = self._l.rsqc_decode_end(self._h, C.byref(info))
        lw = info.last
        self.decode_last = (int(lw.n_records), [int(t) for t in np.ctypeslib.as_array(C.cast(lw.run_tid, C.POINTER(C.c_int32)), (lw.n_runs,))] if lw.n_runs else [])
        names = []
        if info.n_bad_refid and info.bad_refid:
            arr = C.cast(info.bad_refid, C.POINTER(C.c_char_p))
            names = [arr[k].decode() for k in range(min(info.n_bad_refid, 64))]
        self._check(rc)
        return int(info.records), bool(info.unsorted), int(info.n_bad_refid), names

    def read_device(self, ptr, count, dtype):
        """count items of dtype from a device pointer of this context (tests): hipMemcpy of the HIP runtime the library itself is
        linked against, after rsqc_wait."""
        out = np.zeros(count, dtype)
        if count:
            global _hip
            if _hip is None:
                # the very copy of the HIP runtime the product library is running on (a process may hold a second one, e.g.
                # PyTorch's bundled runtime, which knows nothing of this context's allocations)
                path = "libamdhip64.so"
                for line in open("/proc/self/maps"):
                    if "libamdhip64" in line and "/torch/" not in line:
                        path = line.split()[-1]; break
                _hip = C.CDLL(path)
                _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            self.wait()
            rc = _hip.hipMemcpy(out.ctypes.data, ptr, out.nbytes, 2)     # hipMemcpyDeviceToHost
            if rc:
                raise EngineError(abi.ERR_HIP if hasattr(abi, "ERR_HIP") else -1, "hipMemcpy(device -> host) failed: %d" % rc)
        return out

    def last_decoded(self):
        """abi.BatchStruct of DEVICE pointers: rsqc_decode_window.device_batch of the last decode_submit."""
        return self._last_window.device_batch

    def upload(self, batch) -> int:
        s = batch.to_struct()
        h = C.c_int()
        self._check(self._l.rsqc_upload(self._h, C.byref(s), C.byref(h)))
        return h.value

    def upload_struct(self, s) -> int:
        """rsqc_upload of an already packed abi.BatchStruct."""
        h = C.c_int()
        self._check(self._l.rsqc_upload(self._h, C.byref(s), C.byref(h)))
        return h.value

    def submit_resident(self, handle: int):
        self._check(self._l.rsqc_submit_resident(self._h, handle))

    def release(self, handle: int):
        self._check(self._l.rsqc_release(self._h, handle))

    def finalize(self, lazy: bool = False) -> abi.Results:
        """Runs the end-of-file stage and reads every result vector back to the host (inside the library).
        lazy=True returns views that are copied out of the library's host buffers on first access and are only
        valid until the next finalize/reset of this engine."""
        rs = abi.ResultsStruct()
        self._check(self._l.rsqc_finalize(self._h, C.byref(rs)))
        r = abi.Results(rs)
        return r if lazy else r.materialise()

    def finalize_device(self):
        """End-of-file stage without the read-back (the distributed path reduces first, then refresh_results())."""
        self._check(self._l.rsqc_finalize_device(self._h))

    @staticmethod
    def reduce_group(engines) -> bool:
        """rsqc_reduce_group over the contexts of `engines` (all past finalize_device): results summed onto engines[0].
        Returns True when the RCCL path was taken (False: peer copies)."""
        arr = (C.c_void_p * len(engines))(*[e._h for e in engines])
        used = C.c_int(0)
        engines[0]._check(engines[0]._l.rsqc_reduce_group(arr, len(engines), C.byref(used)))
        return bool(used.value)

    class Group:
        """rsqc_group_*: the exchange group of a sharded run, its RCCL communicators made once (any time after the contexts
        exist) and reused by every reduce()."""

        def __init__(self, engines):
            self._engines = list(engines)
            self._l = engines[0]._l
            arr = (C.c_void_p * len(engines))(*[e._h for e in engines])
            h = C.c_void_p()
            engines[0]._check(self._l.rsqc_group_create(arr, len(engines), C.byref(h)))
            self._g = h

        def info(self):
            uses, init_ms, red_ms, note = C.c_int(), C.c_double(), C.c_double(), C.c_char_p()
            self._l.rsqc_group_info(self._g, C.byref(uses), C.byref(init_ms), C.byref(red_ms), C.byref(note))
            return dict(uses_rccl=bool(uses.value), init_ms=init_ms.value, last_reduce_ms=red_ms.value, note=(note.value or b"").decode())

        def reduce(self) -> bool:
            used = C.c_int(0)
            self._engines[0]._check(self._l.rsqc_group_reduce(self._g, C.byref(used)))
            return bool(used.value)

        def close(self):
            if self._g:
                self._l.rsqc_group_destroy(self._g); self._g = None

    def refresh_results(self, lazy: bool = False) -> abi.Results:
        rs = abi.ResultsStruct()
        self._check(self._l.rsqc_refresh_results(self._h, C.byref(rs)))
        r = abi.Results(rs)
        return r if lazy else r.materialise()

    def reset(self):
        self._check(self._l.rsqc_reset(self._h))

    def timing(self) -> dict:
        t = abi.TimingStruct()
        self._check(self._l.rsqc_get_timing(self._h, C.byref(t)))
        return {f: getattr(t, f) for f, _ in abi.TimingStruct._fields_}

    def reset_timing(self):
        self._check(self._l.rsqc_reset_timing(self._h))

    def device_accumulators(self):
        u, f = C.c_void_p(), C.c_void_p()
        nu, nf = C.c_uint64(), C.c_uint64()
        self._check(self._l.rsqc_device_accumulators(self._h, C.byref(u), C.byref(nu), C.byref(f), C.byref(nf)))
        # int64 view: torch has no uint64 arithmetic; counts are far below 2^63
        return DeviceArray(u.value, nu.value, "<i8"), DeviceArray(f.value, nf.value, "<f8")

    def device_vectors(self):
        """The three device ranges a sharded run sum-reduces (rsqc_device_vectors): int64 counts, f64 sums and
        owner-only statistics, u8 validity flags."""
        r = (abi.DeviceRange * 3)()
        self._check(self._l.rsqc_device_vectors(self._h, C.byref(r)))
        return (DeviceArray(r[0].base, r[0].count, "<i8"), DeviceArray(r[1].base, r[1].count, "<f8"),
                DeviceArray(r[2].base, r[2].count, "|u1"))

    def shard_summary(self):
        """The order-dependent outputs of this shard (rsqc_shard_summary) as a distributed.ShardInfo (copies)."""
        from .distributed import ShardInfo
        si = abi.ShardInfoStruct()
        self._check(self._l.rsqc_shard_summary(self._h, C.byref(si)))
        nb, ns = si.n_batches, si.n_samples
        off = abi._view(si.rl_offset, nb + 1, np.uint32).copy() if nb else np.zeros(1, np.uint32)
        ne = int(off[-1])
        return ShardInfo(batch_file_index=abi._view(si.batch_file_index, nb, np.uint64).copy(),
                         batch_records=abi._view(si.batch_records, nb, np.uint64).copy(), rl_offset=off,
                         rl_span=abi._view(si.rl_span, ne, np.uint32).copy(), rl_state=abi._view(si.rl_state, ne, np.int32).copy(),
                         sample_file_index=abi._view(si.sample_file_index, ns, np.uint64).copy(),
                         sample_size=abi._view(si.sample_size, ns, np.uint32).copy())

    def close(self):
        if self._h:
            self._l.rsqc_destroy(self._h)
            self._h = C.c_void_p()
            for p in self._pinned:
                self._l.rsqc_host_free(p)
            self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_engine(params, ann, batches, bed=None, owned=None, reference=None) -> abi.Results:
    e = Engine(params)
    try:
        e.set_annotation(ann, owned)
        if bed is not None:
            e.set_bed(bed)
        if reference is not None:
            e.set_reference(reference)
        for b in batches:
            e.submit(b)
        return e.finalize()
    finally:
        e.close()
