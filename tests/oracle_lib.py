"""ctypes binding of the CPU oracle (oracle/libfl_oracle.so).

Test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this.  The product package never does.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

TYPES = {
    "u8": (np.uint8, 8, ctypes.c_uint8),
    "u16": (np.uint16, 16, ctypes.c_uint16),
    "u32": (np.uint32, 32, ctypes.c_uint32),
    "u64": (np.uint64, 64, ctypes.c_uint64),
}


def tbits(ty):
    return TYPES[ty][1]


def lanes(ty):
    return 1024 // TYPES[ty][1]


def packed_len(ty, w):
    """bitpacking.rs:77  packed_len = 128 * width / size_of::<T>() elements"""
    return 128 * w // (TYPES[ty][1] // 8)


def build_oracle():
    so = os.path.join(ORACLE_DIR, "libfl_oracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("fl_oracle.c", "fl_oracle_impl.inc", "fl_oracle.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so


class Oracle:
    def __init__(self, so):
        self.lib = ctypes.CDLL(so)
        self.lib.fl_oracle_index.restype = ctypes.c_uint
        self.lib.fl_oracle_transpose_index.restype = ctypes.c_uint

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(ctypes.c_void_p)

    def _arr(self, ty, a, n=None):
        a = np.ascontiguousarray(a, dtype=TYPES[ty][0])
        if n is not None:
            assert a.size == n, (a.size, n)
        return a

    def index(self, row, lane):
        return self.lib.fl_oracle_index(row, lane)

    def transpose_index(self, i):
        return self.lib.fl_oracle_transpose_index(i)

    # ---- single-block literal functions -------------------------------------
    def pack(self, ty, w, values):
        v = self._arr(ty, values, 1024)
        out = np.zeros(packed_len(ty, min(w, tbits(ty))), dtype=TYPES[ty][0])
        rc = getattr(self.lib, f"fl_oracle_pack_{ty}")(ctypes.c_uint(w), self._p(v), self._p(out))
        if rc:
            raise ValueError(f"oracle rc={rc}")
        return out

    def unpack(self, ty, w, packed):
        p = self._arr(ty, packed, packed_len(ty, min(w, tbits(ty))))
        out = np.zeros(1024, dtype=TYPES[ty][0])
        rc = getattr(self.lib, f"fl_oracle_unpack_{ty}")(ctypes.c_uint(w), self._p(p), self._p(out))
        if rc:
            raise ValueError(f"oracle rc={rc}")
        return out

    def unpack_single(self, ty, w, packed, index):
        p = self._arr(ty, packed)
        val = TYPES[ty][2](0)
        rc = getattr(self.lib, f"fl_oracle_unpack_single_{ty}")(
            ctypes.c_uint(w), self._p(p), ctypes.c_size_t(index), ctypes.byref(val))
        if rc:
            raise ValueError(f"oracle rc={rc}")
        return val.value

    def for_pack(self, ty, w, values, reference):
        v = self._arr(ty, values, 1024)
        out = np.zeros(packed_len(ty, min(w, tbits(ty))), dtype=TYPES[ty][0])
        rc = getattr(self.lib, f"fl_oracle_for_pack_{ty}")(
            ctypes.c_uint(w), self._p(v), TYPES[ty][2](int(reference)), self._p(out))
        if rc:
            raise ValueError(f"oracle rc={rc}")
        return out

    def unfor_pack(self, ty, w, packed, reference):
        p = self._arr(ty, packed, packed_len(ty, min(w, tbits(ty))))
        out = np.zeros(1024, dtype=TYPES[ty][0])
        rc = getattr(self.lib, f"fl_oracle_unfor_pack_{ty}")(
            ctypes.c_uint(w), self._p(p), TYPES[ty][2](int(reference)), self._p(out))
        if rc:
            raise ValueError(f"oracle rc={rc}")
        return out

    def delta(self, ty, values, base):
        v = self._arr(ty, values, 1024)
        b = self._arr(ty, base, lanes(ty))
        out = np.zeros(1024, dtype=TYPES[ty][0])
        getattr(self.lib, f"fl_oracle_delta_{ty}")(self._p(v), self._p(b), self._p(out))
        return out

    def undelta(self, ty, values, base):
        v = self._arr(ty, values, 1024)
        b = self._arr(ty, base, lanes(ty))
        out = np.zeros(1024, dtype=TYPES[ty][0])
        getattr(self.lib, f"fl_oracle_undelta_{ty}")(self._p(v), self._p(b), self._p(out))
        return out

    def undelta_pack(self, ty, w, packed, base):
        p = self._arr(ty, packed, packed_len(ty, min(w, tbits(ty))))
        b = self._arr(ty, base, lanes(ty))
        out = np.zeros(1024, dtype=TYPES[ty][0])
        rc = getattr(self.lib, f"fl_oracle_undelta_pack_{ty}")(
            ctypes.c_uint(w), self._p(p), self._p(b), self._p(out))
        if rc:
            raise ValueError(f"oracle rc={rc}")
        return out

    def transpose(self, ty, values):
        v = self._arr(ty, values, 1024)
        out = np.zeros(1024, dtype=TYPES[ty][0])
        getattr(self.lib, f"fl_oracle_transpose_{ty}")(self._p(v), self._p(out))
        return out

    def untranspose(self, ty, values):
        v = self._arr(ty, values, 1024)
        out = np.zeros(1024, dtype=TYPES[ty][0])
        getattr(self.lib, f"fl_oracle_untranspose_{ty}")(self._p(v), self._p(out))
        return out

    # ---- batched convenience over the literal functions ---------------------
    def batch(self, op, ty, w, data, aux=None, n_blocks=None):
        """Apply a single-block literal oracle op to n contiguous blocks."""
        dt = TYPES[ty][0]
        data = np.ascontiguousarray(data, dtype=dt)
        pl = packed_len(ty, w) if w is not None else None
        if op in ("pack", "for_pack"):
            n = data.size // 1024
            out = np.zeros(n * pl, dtype=dt)
            for b in range(n):
                blk = data[b * 1024:(b + 1) * 1024]
                out[b * pl:(b + 1) * pl] = (self.pack(ty, w, blk) if op == "pack"
                                            else self.for_pack(ty, w, blk, aux[b]))
            return out
        if op in ("unpack", "unfor_pack", "undelta_pack"):
            n = n_blocks if n_blocks is not None else (data.size // pl if pl else 0)
            out = np.zeros(n * 1024, dtype=dt)
            L = lanes(ty)
            for b in range(n):
                blk = data[b * pl:(b + 1) * pl]
                if op == "unpack":
                    r = self.unpack(ty, w, blk)
                elif op == "unfor_pack":
                    r = self.unfor_pack(ty, w, blk, aux[b])
                else:
                    r = self.undelta_pack(ty, w, blk, aux[b * L:(b + 1) * L])
                out[b * 1024:(b + 1) * 1024] = r
            return out
        if op in ("delta", "undelta"):
            n = data.size // 1024
            out = np.zeros(n * 1024, dtype=dt)
            L = lanes(ty)
            f = self.delta if op == "delta" else self.undelta
            for b in range(n):
                out[b * 1024:(b + 1) * 1024] = f(ty, data[b * 1024:(b + 1) * 1024], aux[b * L:(b + 1) * L])
            return out
        if op in ("transpose", "untranspose"):
            n = data.size // 1024
            out = np.zeros(n * 1024, dtype=dt)
            f = self.transpose if op == "transpose" else self.untranspose
            for b in range(n):
                out[b * 1024:(b + 1) * 1024] = f(ty, data[b * 1024:(b + 1) * 1024])
            return out
        raise KeyError(op)

    def parallel_fill(self, arr, bytes_per_block, n_blocks, seed, nthreads):
        """First-touch `arr` with random bits using the fast family's thread partition."""
        rc = self.lib.fl_oracle_parallel_fill(self._p(arr), ctypes.c_size_t(bytes_per_block),
                                              ctypes.c_size_t(n_blocks), ctypes.c_uint64(seed), ctypes.c_uint(nthreads))
        if rc:
            raise ValueError(f"oracle fill rc={rc}")
        return arr

    def block_hashes(self, ty, v, nthreads=1):
        """(sum, weighted sum) per 1024-value block, wrapping uint64 (full-size content check)."""
        v = np.ascontiguousarray(v, dtype=TYPES[ty][0])
        n = v.size // 1024
        s = np.empty(n, dtype=np.uint64)
        w = np.empty(n, dtype=np.uint64)
        rc = getattr(self.lib, f"fl_oracle_block_hashes_{ty}")(self._p(v), ctypes.c_size_t(n), self._p(s), self._p(w),
                                                               ctypes.c_uint(nthreads))
        if rc:
            raise ValueError(f"oracle hash rc={rc}")
        return s, w

    # ---- fast family (CPU baseline only) -------------------------------------
    def fast(self, op, ty, w, data, aux=None, n_blocks=None, nthreads=1, out=None):
        dt = TYPES[ty][0]
        data = np.ascontiguousarray(data, dtype=dt)
        pl = packed_len(ty, w)
        if op in ("pack", "for_pack"):
            n = data.size // 1024 if n_blocks is None else n_blocks
            out = np.zeros(n * pl, dtype=dt) if out is None else out
        else:
            n = (data.size // pl) if n_blocks is None else n_blocks
            out = np.zeros(n * 1024, dtype=dt) if out is None else out
        fn = getattr(self.lib, f"fl_oracle_fast_{op}_{ty}")
        args = [ctypes.c_uint(w), self._p(data)]
        if op in ("for_pack", "unfor_pack", "undelta_pack"):
            aux = np.ascontiguousarray(aux, dtype=dt)
            args.append(self._p(aux))
        args += [self._p(out), ctypes.c_size_t(n), ctypes.c_uint(nthreads)]
        rc = fn(*args)
        if rc:
            raise ValueError(f"oracle fast rc={rc}")
        return out


    def fast_unpack_mixed_u32(self, widths, offsets, packed, n_blocks=None, nthreads=1, out=None):
        """CPU baseline of BASELINE config 5: the caller loop over per-block widths (bitpacking.rs:109-129)."""
        widths = np.ascontiguousarray(widths, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        packed = np.ascontiguousarray(packed, dtype=np.uint32)
        n = widths.size if n_blocks is None else n_blocks
        out = np.zeros(n * 1024, dtype=np.uint32) if out is None else out
        rc = self.lib.fl_oracle_fast_unpack_mixed_u32(self._p(widths), self._p(offsets), self._p(packed), self._p(out),
                                                      ctypes.c_size_t(n), ctypes.c_uint(nthreads))
        if rc:
            raise ValueError(f"oracle fast mixed rc={rc}")
        return out


def load_native_oracle():
    """-march=native build for the CPU-baseline leg (falls back to the portable build)."""
    try:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "native"])
        return Oracle(os.path.join(ORACLE_DIR, "libfl_oracle_native.so")), "-march=native"
    except Exception:
        return load_oracle(), "-mavx2 -mbmi2 (portable build)"


_ORACLE = None


def load_oracle():
    global _ORACLE
    if _ORACLE is None:
        _ORACLE = Oracle(build_oracle())
    return _ORACLE
