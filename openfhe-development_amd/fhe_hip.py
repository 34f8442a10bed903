"""ctypes binding of the C ABI in include/fhe_hip.h (libfhe_hip.so, built for gfx950).

This is plumbing for tests and bench.py: the product is the C-ABI library itself.  There is NO CPU
fallback: constructing `Lib()` raises if the HIP library has not been built, and every call raises
`FheError` with the library's message if the device is unusable.

Host-side mirror of the reference surface: `Context` ~ ILDCRTParams + twiddle cache, `Tower` ~
lbcrypto::DCRTPoly (src/core/include/lattice/hal/default/dcrtpoly.h:59-398) holding a batch of
device-resident towers, with the reference's method names (SwitchFormat, Plus, Minus, Times,
AutomorphismTransform, ApproxSwitchCRTBasis ...).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_SO = os.path.join(_HERE, "csrc", "libfhe_hip.so")

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
vp = C.c_void_p
u32 = C.c_uint32
u64 = C.c_uint64
EVALUATION, COEFFICIENT = 0, 1  # lbcrypto::Format (src/core/include/utils/inttypes.h:65)


class FheError(RuntimeError):
    pass


def _np_u32(a):
    if a is None:
        return None, None
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.uint32))
    return arr, arr.ctypes.data_as(u32p)


class Lib:
    """Loads libfhe_hip.so (or an explicitly given build, e.g. the test-only lane emulator)."""

    def __init__(self, path=None):
        path = path or os.environ.get("FHE_HIP_LIB") or DEFAULT_SO
        if not os.path.exists(path):
            raise FheError(
                f"{path} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback.")
        self.path = path
        L = self.L = C.CDLL(path)

        def S(name, res, args):
            f = getattr(L, name)
            f.restype, f.argtypes = res, args

        S("fhe_last_error", C.c_char_p, [])
        S("fhe_version", C.c_char_p, [])
        S("fhe_device_count", C.c_int, [])
        S("fhe_ctx_create", C.c_int, [u32, u32, u64p, u64p, C.c_int, C.POINTER(vp)])
        S("fhe_ctx_destroy", None, [vp])
        S("fhe_ctx_logn", u32, [vp])
        S("fhe_ctx_limbs", u32, [vp])
        S("fhe_ctx_device", C.c_int, [vp])
        S("fhe_malloc", C.c_int, [vp, C.c_size_t, C.POINTER(vp)])
        S("fhe_free", C.c_int, [vp, vp])
        for n in ("fhe_memcpy_h2d", "fhe_memcpy_d2h", "fhe_memcpy_d2d"):
            S(n, C.c_int, [vp, vp, vp, C.c_size_t, vp])
        S("fhe_stream_sync", C.c_int, [vp, vp])
        S("fhe_stream_create", C.c_int, [vp, C.POINTER(vp)])
        S("fhe_stream_destroy", C.c_int, [vp, vp])
        S("fhe_stream_wait", C.c_int, [vp, vp, vp])
        S("fhe_memset_zero", C.c_int, [vp, vp, C.c_size_t, vp])
        S("fhe_checksum", C.c_int, [vp, vp, u32, vp, vp])
        S("fhe_graph_begin", C.c_int, [vp, vp])
        S("fhe_graph_end", C.c_int, [vp, vp, C.POINTER(vp)])
        S("fhe_graph_launch", C.c_int, [vp, vp, vp])
        S("fhe_graph_destroy", None, [vp])
        S("fhe_ntt_fwd", C.c_int, [vp, vp, u32p, u32, u32, vp])
        S("fhe_ntt_inv", C.c_int, [vp, vp, u32p, u32, u32, vp])
        S("fhe_ntt_fwd_oop", C.c_int, [vp, vp, vp, u32p, u32, u32, vp])
        S("fhe_ntt_inv_oop", C.c_int, [vp, vp, vp, u32p, u32, u32, vp])
        for n in ("fhe_add", "fhe_sub", "fhe_mul", "fhe_mul_add"):
            S(n, C.c_int, [vp, vp, vp, vp, u32p, u32, u32, vp])
        S("fhe_neg", C.c_int, [vp, vp, vp, u32p, u32, u32, vp])
        S("fhe_mul_const", C.c_int, [vp, vp, vp, u64p, u32p, u32, u32, vp])
        S("fhe_mult_acc", C.c_int, [vp, vp, vp, u64p, u32p, u32, u32, vp])
        S("fhe_times_q_over_t", C.c_int, [vp, vp, vp, u64, u64, u64p, u32p, u32, u32, vp])
        S("fhe_mod_switch_round", C.c_int, [vp, vp, u64, u64, vp, C.c_size_t, vp])
        S("fhe_inner_product", C.c_int, [vp, u32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), u32p, u32p, u32, u32, vp, vp, vp])
        S("fhe_add_const", C.c_int, [vp, vp, vp, u64p, u32p, u32, u32, C.c_int, vp])
        S("fhe_sub_const", C.c_int, [vp, vp, vp, u64p, u32p, u32, u32, vp])
        S("fhe_poly_mul_workspace_bytes", C.c_size_t, [vp, u32, u32])
        S("fhe_poly_mul", C.c_int, [vp, vp, vp, vp, u32p, u32, u32, vp, C.c_size_t, vp])
        S("fhe_tensor_square", C.c_int, [vp, vp, vp, vp, vp, vp, u32p, u32, u32, vp])
        S("fhe_mod_up", C.c_int, [vp, vp, C.c_int, vp, u32, vp, C.c_size_t, vp])
        S("fhe_expand_crt_basis_ql_hat", C.c_int, [vp, vp, u32, u64p, u32p, u32, u32, vp, vp])
        S("fhe_tensor", C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, u32p, u32, u32, vp])
        S("fhe_automorph", C.c_int, [vp, vp, vp, u32, C.c_int, u32p, u32, u32, vp])
        S("fhe_switch_modulus", C.c_int, [vp, vp, u32p, u32, vp, u32, u32, u32, u32, vp])
        S("fhe_crt_decompose_towers", u32, [vp, u32p, u32, u32])
        S("fhe_crt_decompose", C.c_int, [vp, vp, u32p, u32, u32, vp, vp])
        S("fhe_event_create", C.c_int, [vp, C.POINTER(vp)])
        S("fhe_event_record", C.c_int, [vp, vp, vp])
        S("fhe_stream_wait_event", C.c_int, [vp, vp, vp])
        S("fhe_event_destroy", C.c_int, [vp, vp])
        S("fhe_sample_uniform", C.c_int, [vp, vp, u32p, u32, u32, C.c_uint64, u32, vp])
        S("fhe_sample_gaussian", C.c_int, [vp, vp, u32p, u32, u32, C.c_double, C.c_uint64, u32, vp])
        S("fhe_sample_ternary", C.c_int, [vp, vp, u32p, u32, u32, C.c_uint64, u32, vp])
        S("fhe_conv_create", C.c_int, [vp, u32p, u32, u32p, u32, C.POINTER(vp)])
        S("fhe_conv_destroy", None, [vp])
        S("fhe_approx_switch_basis", C.c_int, [vp, vp, u32, u32, vp, u32, u32, u32, vp])
        S("fhe_switch_basis_exact", C.c_int, [vp, vp, u32, u32, vp, u32, u32, u32, vp])
        S("fhe_conv_create_custom", C.c_int, [vp, u32p, u32, u32p, u32, u64p, u64p, u64p, C.POINTER(C.c_double), C.POINTER(vp)])
        S("fhe_expand_crt_basis_workspace_bytes", C.c_size_t, [vp, u32])
        S("fhe_expand_crt_basis", C.c_int, [vp, vp, C.c_int, vp, C.c_int, C.c_int, u32, vp, C.c_size_t, vp])
        S("fhe_fast_expand_crt_basis_p_over_q", C.c_int, [vp, vp, vp, vp, u32, vp])
        S("fhe_ks_plan_create", C.c_int, [vp, u32, u32, u32, C.POINTER(vp)])
        S("fhe_ks_plan_destroy", None, [vp])
        S("fhe_ks_plan_alpha", u32, [vp])
        S("fhe_ks_key_alloc", C.c_int, [vp, C.POINTER(vp)])
        S("fhe_ks_key_upload", C.c_int, [vp, u64p, u64p, C.POINTER(vp)])
        S("fhe_ks_key_wrap", C.c_int, [vp, vp, vp, C.POINTER(vp)])
        S("fhe_ks_key_destroy", None, [vp])
        S("fhe_ks_key_devptr", vp, [vp, C.c_int])
        S("fhe_ks_key_words", C.c_size_t, [vp])
        S("fhe_ks_workspace_bytes", C.c_size_t, [vp, u32, u32])
        S("fhe_keyswitch_hybrid", C.c_int, [vp, vp, vp, u32, u32, vp, vp, vp, C.c_size_t, vp])
        S("fhe_keyswitch_hybrid_acc", C.c_int, [vp, vp, vp, u32, u32, vp, vp, vp, C.c_size_t, vp])
        S("fhe_ckks_eval_mult", C.c_int, [vp, vp, vp, vp, vp, vp, u32, u32, vp, vp, vp, C.c_size_t, vp])
        S("fhe_ks_precompute", C.c_int, [vp, vp, u32, u32, vp, C.c_size_t, vp])
        S("fhe_ks_fast_keyswitch", C.c_int, [vp, vp, vp, u32, u32, vp, vp, vp, C.c_size_t, vp])
        S("fhe_eval_fast_rotation", C.c_int, [vp, vp, vp, vp, u32, u32, u32, vp, vp, vp, C.c_size_t, vp])
        S("fhe_eval_automorphism", C.c_int, [vp, vp, vp, vp, u32, u32, u32, vp, vp, vp, C.c_size_t, vp])
        S("fhe_ks_ext", C.c_int, [vp, vp, u32, u32, vp, vp])
        S("fhe_ks_fast_keyswitch_ext", C.c_int, [vp, vp, vp, u32, u32, vp, vp, vp, C.c_size_t, vp])
        S("fhe_eval_fast_rotation_ext", C.c_int, [vp, vp, vp, vp, u32, C.c_int, u32, u32, vp, vp, vp, C.c_size_t, vp])
        S("fhe_ks_down", C.c_int, [vp, vp, vp, u32, u32, vp, vp, vp, C.c_size_t, vp])
        S("fhe_ckks_bsgs_workspace_bytes", C.c_size_t, [vp, u32, u32, u32, u32])
        S("fhe_ckks_bsgs_transform", C.c_int, [vp, vp, vp, u32, u32, u32, vp, vp, u32, vp, vp, vp, vp, vp, vp, C.c_size_t, vp])
        S("fhe_approx_mod_down", C.c_int, [vp, vp, u32, u32, vp, vp, C.c_size_t, vp])
        S("fhe_approx_mod_down_bgv", C.c_int, [vp, vp, u32, u64, u32, vp, vp, C.c_size_t, vp])
        S("fhe_rescale_workspace_bytes", C.c_size_t, [vp, u32, u32])
        S("fhe_rescale", C.c_int, [vp, vp, u32, u32, vp, vp, C.c_size_t, vp])
        S("fhe_rescale_limbs", C.c_int, [vp, vp, u32p, u32, u64p, u64p, u32, vp, vp, C.c_size_t, vp])
        S("fhe_rescale_limbs_pair", C.c_int, [vp, vp, vp, u32p, u32, u64p, u64p, vp, vp, vp, C.c_size_t, vp])
        S("fhe_add_pair", C.c_int, [vp, vp, vp, vp, vp, vp, vp, u32p, u32, vp])
        S("fhe_sub_pair", C.c_int, [vp, vp, vp, vp, vp, vp, vp, u32p, u32, vp])
        S("fhe_mul_const_pair", C.c_int, [vp, vp, vp, vp, vp, u64p, u32p, u32, vp])
        S("fhe_mem_info", C.c_int, [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)])
        S("fhe_lincomb", C.c_int, [vp, vp, C.POINTER(vp), u64p, u32, u32p, u32, u32, C.c_int, vp])
        S("fhe_mod_reduce", C.c_int, [vp, vp, u32, u64, C.c_int, u32, vp, vp, C.c_size_t, vp])
        f64p = C.POINTER(C.c_double)
        S("fhe_sr_plan_create", C.c_int, [vp, u32, u32p, u32, u64p, f64p, C.POINTER(vp)])
        S("fhe_sr_plan_destroy", None, [vp])
        S("fhe_scale_and_round", C.c_int, [vp, vp, C.c_int, vp, u32, vp])
        S("fhe_scale_and_round_p_over_q", C.c_int, [vp, vp, u32p, u32, vp, u32, vp])
        S("fhe_scale_and_round_native", C.c_int, [vp, vp, u32p, u32, u64, u64p, u64p, f64p, f64p, u32, vp, vp])
        S("fhe_scale_and_round_behz_decrypt", C.c_int, [vp, vp, u32p, u32, u64, u64p, u64p, u32, vp, vp])
        S("fhe_param_behz_bsk", u32, [u32, u32, u64p, u64, u64p, u64p])
        S("fhe_behz_create", C.c_int, [vp, u32p, u32, u32p, u64, C.POINTER(vp)])
        S("fhe_behz_destroy", None, [vp])
        S("fhe_behz_override_q_to_bsk", C.c_int, [vp, u64p, u64p, u64p, u64p, u64, u64p])
        S("fhe_behz_override_floorq", C.c_int, [vp, u64p, u64p, u64p, u64p])
        S("fhe_behz_override_conv_sk", C.c_int, [vp, u64p, u64p, u64, u64p, u64p])
        S("fhe_behz_workspace_bytes", C.c_size_t, [vp, u32])
        S("fhe_behz_q_to_bsk", C.c_int, [vp, vp, C.c_int, u32, vp, C.c_size_t, vp])
        S("fhe_behz_floorq", C.c_int, [vp, vp, u32, vp])
        S("fhe_behz_conv_sk", C.c_int, [vp, vp, vp, u32, vp])
        S("fhe_bfv_eval_mult_behz_workspace_bytes", C.c_size_t, [vp, u32])
        S("fhe_bfv_eval_mult_behz", C.c_int, [vp] * 8 + [C.c_int, u32, vp, C.c_size_t, vp])
        S("fhe_bfv_eval_mult_relin_workspace_bytes", C.c_size_t, [vp, vp, u32])
        S("fhe_bfv_eval_mult_relin_behz", C.c_int, [vp] * 9 + [u32, vp, C.c_size_t, vp])
        S("fhe_param_first_prime", u64, [u32, u64])
        S("fhe_param_last_prime", u64, [u32, u64])
        S("fhe_param_next_prime", u64, [u64, u64])
        S("fhe_param_previous_prime", u64, [u64, u64])
        S("fhe_param_root_of_unity", u64, [u64, u64])
        S("fhe_param_dcrt_chain", C.c_int, [u32, u32, u32, u64p, u64p])
        S("fhe_param_select_p", u32, [u32, u32, u64p, u32, u32, u64p, u64p])
        S("fhe_param_find_automorphism_index_2n_complex", u32, [C.c_int32, u32])
        S("fhe_launch_stats", C.c_size_t, [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64)])
        S("fhe_time_ntt", C.c_int, [vp, vp, u32p, u32, u32, C.c_int, C.c_int, vp, C.POINTER(C.c_float)])

    def check(self, status):
        if status != 0:
            raise FheError(f"fhe status {status}: {self.L.fhe_last_error().decode()}")

    def launch_count(self, kernel):
        """launches of `kernel` (name without template arguments) since the library was loaded (fhe_launch_stats)"""
        buf = C.create_string_buffer(1 << 16)
        self.L.fhe_launch_stats(buf, len(buf), None)
        for line in buf.value.decode().splitlines():
            k, n = line.rsplit(" ", 1)
            if k.split("::")[-1] == kernel:
                return int(n)
        return 0

    def version(self):
        return self.L.fhe_version().decode()

    def device_count(self):
        return self.L.fhe_device_count()

    # ---- host-side parameter helpers (ILDCRTParams chain, HYBRID auxiliary basis) ----
    def dcrt_chain(self, logN, n_limbs, bits):
        q = np.zeros(n_limbs, np.uint64)
        psi = np.zeros(n_limbs, np.uint64)
        self.check(self.L.fhe_param_dcrt_chain(2 << logN, n_limbs, bits, q.ctypes.data_as(u64p), psi.ctypes.data_as(u64p)))
        return q, psi

    def ckks_like_chain(self, logN, sizeQ, first_bits=60, scale_bits=59):
        """first modulus of first_bits, then sizeQ-1 descending primes of scale_bits (FIXEDMANUAL-like shape)"""
        M = 2 << logN
        q = [self.L.fhe_param_last_prime(first_bits, M)]
        cur = self.L.fhe_param_last_prime(scale_bits, M)
        while len(q) < sizeQ:
            if cur not in q:
                q.append(cur)
            cur = self.L.fhe_param_previous_prime(cur, M)
        q = np.array(q, np.uint64)
        psi = np.array([self.L.fhe_param_root_of_unity(M, int(v)) for v in q], np.uint64)
        return q, psi

    def behz_bsk(self, logN, q, t):
        q = np.ascontiguousarray(q, dtype=np.uint64)
        bsk = np.zeros(len(q) + 1, np.uint64)
        psi = np.zeros(len(q) + 1, np.uint64)
        n = self.L.fhe_param_behz_bsk(logN, len(q), q.ctypes.data_as(u64p), t, bsk.ctypes.data_as(u64p), psi.ctypes.data_as(u64p))
        if n == 0:
            raise FheError("fhe_param_behz_bsk failed")
        return bsk, psi

    def find_automorphism_index(self, index, m):
        """FindAutomorphismIndex2nComplex (nbtheory2.cpp:243-262)"""
        k = self.L.fhe_param_find_automorphism_index_2n_complex(index, m)
        if k == 0:
            raise FheError("m should be a power of two.")
        return k

    def select_p(self, logN, q, numPartQ, aux_bits=60):
        q = np.ascontiguousarray(q, dtype=np.uint64)
        p = np.zeros(64, np.uint64)
        psi = np.zeros(64, np.uint64)
        n = self.L.fhe_param_select_p(logN, len(q), q.ctypes.data_as(u64p), numPartQ, aux_bits, p.ctypes.data_as(u64p),
                                      psi.ctypes.data_as(u64p))
        if n == 0:
            raise FheError("fhe_param_select_p failed")
        return p[:n].copy(), psi[:n].copy()


class Context:
    """Ring dimension N = 2^logN + modulus tower (q_i, psi_i) with device-resident twiddle tables."""

    def __init__(self, lib, logN, q, psi, device=0):
        self.lib, self.logN, self.N = lib, logN, 1 << logN
        self.q = np.ascontiguousarray(np.asarray(q, dtype=np.uint64))
        self.psi = np.ascontiguousarray(np.asarray(psi, dtype=np.uint64))
        self.L = len(self.q)
        h = vp()
        lib.check(lib.L.fhe_ctx_create(logN, self.L, self.q.ctypes.data_as(u64p), self.psi.ctypes.data_as(u64p),
                                       device, C.byref(h)))
        self.h = h
        self._allocs = {}  # pointer value -> handle

    def close(self):
        if self.h:
            for p in self._allocs.values():
                self.lib.L.fhe_free(self.h, p)
            self._allocs = {}
            self.lib.L.fhe_ctx_destroy(self.h)
            self.h = None

    # ---- raw device memory ----
    def malloc(self, nbytes):
        p = vp()
        self.lib.check(self.lib.L.fhe_malloc(self.h, nbytes, C.byref(p)))
        self._allocs[p.value] = p
        return p

    def free(self, p):
        if self._allocs.pop(p.value, None) is not None:
            self.lib.check(self.lib.L.fhe_free(self.h, p))

    def upload(self, arr, stream=None):
        arr = np.ascontiguousarray(arr, dtype=np.uint64)
        p = self.malloc(arr.nbytes)
        self.lib.check(self.lib.L.fhe_memcpy_h2d(self.h, p, arr.ctypes.data_as(vp), arr.nbytes, stream))
        self.sync(stream)
        return p

    def download(self, p, shape, stream=None):
        out = np.empty(shape, dtype=np.uint64)
        self.lib.check(self.lib.L.fhe_memcpy_d2h(self.h, out.ctypes.data_as(vp), p, out.nbytes, stream))
        self.sync(stream)
        return out

    def sync(self, stream=None):
        self.lib.check(self.lib.L.fhe_stream_sync(self.h, stream))

    def checksum(self, tower, stream=None):
        """uint64[batch * limbs][2] = {sum, position-weighted sum} mod 2^64 of every limb-row of the tower (fhe_checksum)"""
        rows = tower.batch * tower.n_limbs
        d = self.malloc(rows * 16)
        try:
            self.lib.check(self.lib.L.fhe_checksum(self.h, tower.ptr, rows, d, stream))
            return self.download(d, (rows, 2), stream)
        finally:
            self.free(d)

    def tower(self, host, limb_idx=None, fmt=EVALUATION):
        """host: uint64 [batch][nLimbs][N] (or [nLimbs][N])"""
        host = np.asarray(host, dtype=np.uint64)
        if host.ndim == 2:
            host = host[None]
        return Tower(self, self.upload(host), host.shape[0], host.shape[1], limb_idx, fmt, owned=True)

    def empty(self, batch, n_limbs, limb_idx=None, fmt=EVALUATION):
        return Tower(self, self.malloc(batch * n_limbs * self.N * 8), batch, n_limbs, limb_idx, fmt, owned=True)

    # ---- sampled towers (the sampling constructors of DCRTPolyImpl, dcrtpoly-impl.h:126-205, on the device: fhe_sample_*) ----
    def sample(self, kind, batch, n_limbs, seed, stream_id, limb_idx=None, sigma=3.19, stream=None):
        """kind: "uniform" (DugType), "gaussian" (DggType, standard deviation sigma) or "ternary" (TugType); COEFFICIENT format"""
        t = self.empty(batch, n_limbs, limb_idx, COEFFICIENT)
        L = self.lib.L
        if kind == "uniform":
            self.lib.check(L.fhe_sample_uniform(self.h, t.ptr, t._li(), n_limbs, batch, seed, stream_id, stream))
        elif kind == "gaussian":
            self.lib.check(L.fhe_sample_gaussian(self.h, t.ptr, t._li(), n_limbs, batch, sigma, seed, stream_id, stream))
        elif kind == "ternary":
            self.lib.check(L.fhe_sample_ternary(self.h, t.ptr, t._li(), n_limbs, batch, seed, stream_id, stream))
        else:
            raise ValueError(kind)
        return t


class Tower:
    """A batch of device-resident RNS towers: the DCRTPoly data model, uint64[batch][nLimbs][N]."""

    def __init__(self, ctx, ptr, batch, n_limbs, limb_idx=None, fmt=EVALUATION, owned=False):
        self.ctx, self.ptr, self.batch, self.n_limbs, self.fmt = ctx, ptr, batch, n_limbs, fmt
        self._owned = owned  # allocated by Context.tower()/empty(): freed with the object
        self.limb_idx = None if limb_idx is None else np.ascontiguousarray(np.asarray(limb_idx, dtype=np.uint32))

    def _li(self):
        return None if self.limb_idx is None else self.limb_idx.ctypes.data_as(u32p)

    def to_host(self):
        return self.ctx.download(self.ptr, (self.batch, self.n_limbs, self.ctx.N))

    def like(self):
        return self.ctx.empty(self.batch, self.n_limbs, self.limb_idx, self.fmt)

    def free(self):
        """release the device buffer now (a Tower that is simply dropped is released by the garbage collector)"""
        if self.ptr is not None and self.ctx.h:
            self.ctx.free(self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            if getattr(self, "_owned", False):
                self.free()
        except Exception:
            pass

    def MultAccEqNoCheck(self, v, consts, stream=None):  # poly.h:323 / mubintvecnat.cpp:132-142, per limb
        consts = np.ascontiguousarray(np.asarray(consts, dtype=np.uint64))
        self.ctx.lib.check(self.ctx.lib.L.fhe_mult_acc(self.ctx.h, self.ptr, v.ptr, consts.ctypes.data_as(u64p), self._li(),
                                                      self.n_limbs, self.batch, stream))
        return self

    # DCRTPolyImpl::SwitchFormat (dcrtpoly-impl.h:1932-1940)
    def SwitchFormat(self, stream=None):
        L, c = self.ctx.lib, self.ctx
        f = L.L.fhe_ntt_inv if self.fmt == EVALUATION else L.L.fhe_ntt_fwd
        L.check(f(c.h, self.ptr, self._li(), self.n_limbs, self.batch, stream))
        self.fmt = COEFFICIENT if self.fmt == EVALUATION else EVALUATION
        return self

    def SetFormat(self, fmt, stream=None):  # ilelement.h:447-450
        if fmt != self.fmt:
            self.SwitchFormat(stream)
        return self

    # DCRTPolyImpl::CRTDecompose(baseBits) (dcrtpoly-impl.h:230-285): the towers of KeySwitchBV's digit decomposition, EVALUATION format,
    # as ONE batch [towers][nLimbs][N] (batch must be 1; None when the digit size is outside the device path)
    def CRTDecompose(self, base_bits, stream=None):
        assert self.batch == 1
        L, c = self.ctx.lib, self.ctx
        towers = L.L.fhe_crt_decompose_towers(c.h, self._li(), self.n_limbs, base_bits)
        if towers == 0:
            return None
        src = self
        if self.fmt == EVALUATION:  # (:231-233: the coefficient copy)
            src = c.empty(1, self.n_limbs, self.limb_idx, COEFFICIENT)
            L.check(L.L.fhe_ntt_inv_oop(c.h, self.ptr, src.ptr, self._li(), self.n_limbs, 1, stream))
        out = c.empty(towers, self.n_limbs, self.limb_idx, EVALUATION)
        L.check(L.L.fhe_crt_decompose(c.h, src.ptr, self._li(), self.n_limbs, base_bits, out.ptr, stream))
        return out

    def _bin(self, fn, other, stream):
        out = self.like()
        self.ctx.lib.check(fn(self.ctx.h, out.ptr, self.ptr, other.ptr, self._li(), self.n_limbs, self.batch, stream))
        return out

    def Plus(self, other, stream=None):  # dcrtpoly.h:153-162
        return self._bin(self.ctx.lib.L.fhe_add, other, stream)

    def Minus(self, other, stream=None):  # dcrtpoly-impl.h:362-371
        return self._bin(self.ctx.lib.L.fhe_sub, other, stream)

    def Times(self, other, stream=None):  # dcrtpoly.h:174-189
        if isinstance(other, Tower):
            return self._bin(self.ctx.lib.L.fhe_mul, other, stream)
        consts = np.ascontiguousarray(np.asarray(other, dtype=np.uint64))  # Times(vector<NativeInteger>) :582-601
        out = self.like()
        self.ctx.lib.check(self.ctx.lib.L.fhe_mul_const(self.ctx.h, out.ptr, self.ptr, consts.ctypes.data_as(u64p),
                                                       self._li(), self.n_limbs, self.batch, stream))
        return out

    def PolyMul(self, other, stream=None):
        """negacyclic product of two COEFFICIENT towers (fhe_poly_mul): INTT(NTT(a) o NTT(b)) per limb"""
        L = self.ctx.lib.L
        out = self.like()
        wsb = L.fhe_poly_mul_workspace_bytes(self.ctx.h, self.n_limbs, self.batch)
        ws = self.ctx.malloc(wsb)
        try:
            self.ctx.lib.check(L.fhe_poly_mul(self.ctx.h, self.ptr, other.ptr, out.ptr, self._li(), self.n_limbs, self.batch, ws, wsb, stream))
            self.ctx.sync(stream)
        finally:
            self.ctx.free(ws)
        return out

    def Negate(self, stream=None):  # dcrtpoly-impl.h:347-354
        out = self.like()
        self.ctx.lib.check(self.ctx.lib.L.fhe_neg(self.ctx.h, out.ptr, self.ptr, self._li(), self.n_limbs, self.batch,
                                                  stream))
        return out

    def AutomorphismTransform(self, k, stream=None):  # dcrtpoly-impl.h:314-333
        out = self.like()
        self.ctx.lib.check(self.ctx.lib.L.fhe_automorph(self.ctx.h, out.ptr, self.ptr, k,
                                                        1 if self.fmt == EVALUATION else 0, self._li(), self.n_limbs,
                                                        self.batch, stream))
        return out


class Conv:
    """CRT basis conversion plan (ApproxSwitchCRTBasis / SwitchCRTBasis)."""

    def __init__(self, ctx, src_idx, dst_idx, hat_inv=None, hat_mod=None, alpha_mod=None, q_inv=None):
        """default: the plain CRT tables of (src basis, dst basis); with hat_inv[nSrc] / hat_mod[nSrc][nDst]
        (+ alpha_mod[nSrc+1][nDst], q_inv[nSrc] for the exact variant) the caller's tables (fhe_conv_create_custom)"""
        self.ctx = ctx
        self.src = np.ascontiguousarray(np.asarray(src_idx, dtype=np.uint32))
        self.dst = np.ascontiguousarray(np.asarray(dst_idx, dtype=np.uint32))
        h = vp()
        if hat_inv is None:
            ctx.lib.check(ctx.lib.L.fhe_conv_create(ctx.h, self.src.ctypes.data_as(u32p), len(self.src),
                                                    self.dst.ctypes.data_as(u32p), len(self.dst), C.byref(h)))
        else:
            hi = np.ascontiguousarray(hat_inv, dtype=np.uint64)
            hm = np.ascontiguousarray(hat_mod, dtype=np.uint64)
            assert hm.shape == (len(self.src), len(self.dst))
            al = qi = None
            if alpha_mod is not None:
                al_a = np.ascontiguousarray(alpha_mod, dtype=np.uint64)
                qi_a = np.ascontiguousarray(q_inv, dtype=np.float64)
                al, qi = al_a.ctypes.data_as(u64p), qi_a.ctypes.data_as(C.POINTER(C.c_double))
            ctx.lib.check(ctx.lib.L.fhe_conv_create_custom(ctx.h, self.src.ctypes.data_as(u32p), len(self.src),
                                                           self.dst.ctypes.data_as(u32p), len(self.dst),
                                                           hi.ctypes.data_as(u64p), hm.ctypes.data_as(u64p), al, qi,
                                                           C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.ctx.lib.L.fhe_conv_destroy(self.h)
            self.h = None

    def run(self, tin, exact=False, stream=None):
        """tin: Tower [batch][nSrc][N] COEFFICIENT -> Tower [batch][nDst][N] COEFFICIENT"""
        out = self.ctx.empty(tin.batch, len(self.dst), self.dst, COEFFICIENT)
        f = self.ctx.lib.L.fhe_switch_basis_exact if exact else self.ctx.lib.L.fhe_approx_switch_basis
        self.ctx.lib.check(f(self.h, tin.ptr, tin.n_limbs, 0, out.ptr, len(self.dst), 0, tin.batch, stream))
        return out

    def ExpandCRTBasis(self, tin, result_format, reverse=False, stream=None):
        """DCRTPoly::ExpandCRTBasis[ReverseOrder] (dcrtpoly-impl.h:1088-1148): tower over the source basis -> Q u P"""
        idx = np.concatenate([self.dst, self.src]) if reverse else np.concatenate([self.src, self.dst])
        out = self.ctx.empty(tin.batch, len(idx), idx, result_format)
        L = self.ctx.lib.L
        wsb = L.fhe_expand_crt_basis_workspace_bytes(self.h, tin.batch)
        ws = self.ctx.malloc(wsb)
        try:
            self.ctx.lib.check(L.fhe_expand_crt_basis(self.h, tin.ptr, 1 if tin.fmt == EVALUATION else 0, out.ptr,
                                                      1 if result_format == EVALUATION else 0, 1 if reverse else 0,
                                                      tin.batch, ws, wsb, stream))
            self.ctx.sync(stream)
        finally:
            self.ctx.free(ws)
        return out

    def ApproxModUp(self, tin, stream=None):
        """DCRTPoly::ApproxModUp (dcrtpoly-impl.h:935-963): tower over the source basis (either format) -> Q u P, EVALUATION"""
        idx = np.concatenate([self.src, self.dst])
        out = self.ctx.empty(tin.batch, len(idx), idx, EVALUATION)
        L = self.ctx.lib.L
        wsb = L.fhe_expand_crt_basis_workspace_bytes(self.h, tin.batch)
        ws = self.ctx.malloc(wsb)
        try:
            self.ctx.lib.check(L.fhe_mod_up(self.h, tin.ptr, 1 if tin.fmt == EVALUATION else 0, out.ptr, tin.batch, ws, wsb, stream))
            self.ctx.sync(stream)
        finally:
            self.ctx.free(ws)
        return out

    def FastExpandCRTBasisPloverQ(self, to_ql, tin, stream=None):
        """self = custom-table plan Q -> Pl, to_ql = plan Pl -> Ql (dcrtpoly-impl.h:1151-1164); COEFFICIENT towers"""
        idx = np.concatenate([to_ql.dst, self.dst])
        out = self.ctx.empty(tin.batch, len(idx), idx, COEFFICIENT)
        self.ctx.lib.check(self.ctx.lib.L.fhe_fast_expand_crt_basis_p_over_q(self.h, to_ql.h, tin.ptr, out.ptr, tin.batch, stream))
        return out


class KeySwitchPlan:
    """HYBRID key switching for a context whose limbs are Q (sizeQ) followed by P (sizeP)."""

    def __init__(self, ctx, sizeQ, sizeP, numPartQ):
        self.ctx, self.sizeQ, self.sizeP, self.numPartQ = ctx, sizeQ, sizeP, numPartQ
        h = vp()
        ctx.lib.check(ctx.lib.L.fhe_ks_plan_create(ctx.h, sizeQ, sizeP, numPartQ, C.byref(h)))
        self.h = h
        self.key = None
        self._ws = None
        self._ws_bytes = 0

    def close(self):
        if self.key:
            self.ctx.lib.L.fhe_ks_key_destroy(self.key)
            self.key = None
        if self.h:
            self.ctx.lib.L.fhe_ks_plan_destroy(self.h)
            self.h = None

    def upload_key(self, keyB, keyA):
        keyB = np.ascontiguousarray(keyB, dtype=np.uint64)
        keyA = np.ascontiguousarray(keyA, dtype=np.uint64)
        k = vp()
        self.ctx.lib.check(self.ctx.lib.L.fhe_ks_key_upload(self.h, keyB.ctypes.data_as(u64p),
                                                            keyA.ctypes.data_as(u64p), C.byref(k)))
        self.key = k

    def wrap_key(self, dev_b, dev_a):
        """adopt device-resident key vectors (e.g. torch tensors filled by an RCCL broadcast)"""
        k = vp()
        self.ctx.lib.check(self.ctx.lib.L.fhe_ks_key_wrap(self.h, vp(dev_b), vp(dev_a), C.byref(k)))
        self.key = k

    def key_words(self):
        N = self.ctx.N
        return self.numPartQ * (self.sizeQ + self.sizeP) * N

    def workspace(self, sizeQl, batch):
        need = self.ctx.lib.L.fhe_ks_workspace_bytes(self.h, sizeQl, batch)
        if need > self._ws_bytes:
            if self._ws is not None:
                self.ctx.free(self._ws)
            self._ws = self.ctx.malloc(need)
            self._ws_bytes = need
        return self._ws, self._ws_bytes

    def KeySwitchCore(self, c, stream=None):  # keyswitch-hybrid.cpp:308-312
        ws, wsb = self.workspace(c.n_limbs, c.batch)
        o0, o1 = c.like(), c.like()
        self.ctx.lib.check(self.ctx.lib.L.fhe_keyswitch_hybrid(self.h, self.key, c.ptr, c.n_limbs, c.batch, o0.ptr,
                                                               o1.ptr, ws, wsb, stream))
        return o0, o1

    def KeySwitchCoreAcc(self, c, acc0, acc1, stream=None):  # base-leveledshe.cpp:207-211: acc += KeySwitchCore(c), in place
        ws, wsb = self.workspace(c.n_limbs, c.batch)
        self.ctx.lib.check(self.ctx.lib.L.fhe_keyswitch_hybrid_acc(self.h, self.key, c.ptr, c.n_limbs, c.batch, acc0.ptr, acc1.ptr, ws, wsb,
                                                                   stream))

    def EvalMult(self, a0, a1, b0, b1, stream=None):  # base-leveledshe.cpp:201-214
        ws, wsb = self.workspace(a0.n_limbs, a0.batch)
        c0, c1 = a0.like(), a0.like()
        self.ctx.lib.check(self.ctx.lib.L.fhe_ckks_eval_mult(self.h, self.key, a0.ptr, a1.ptr, b0.ptr, b1.ptr,
                                                             a0.n_limbs, a0.batch, c0.ptr, c1.ptr, ws, wsb, stream))
        return c0, c1

    def make_key(self, keyB, keyA):
        """upload an additional evaluation key (e.g. a rotation key); returns the handle"""
        keyB = np.ascontiguousarray(keyB, dtype=np.uint64)
        keyA = np.ascontiguousarray(keyA, dtype=np.uint64)
        k = vp()
        self.ctx.lib.check(self.ctx.lib.L.fhe_ks_key_upload(self.h, keyB.ctypes.data_as(u64p), keyA.ctypes.data_as(u64p),
                                                            C.byref(k)))
        return k

    def EvalFastRotationPrecompute(self, c1, stream=None):  # base-leveledshe.cpp:425-430
        ws, wsb = self.workspace(c1.n_limbs, c1.batch)
        self.ctx.lib.check(self.ctx.lib.L.fhe_ks_precompute(self.h, c1.ptr, c1.n_limbs, c1.batch, ws, wsb, stream))

    def EvalFastRotation(self, key, c0, c1, k, stream=None):  # base-leveledshe.cpp:432-463 (digits already in ws)
        ws, wsb = self.workspace(c0.n_limbs, c0.batch)
        o0, o1 = c0.like(), c0.like()
        self.ctx.lib.check(self.ctx.lib.L.fhe_eval_fast_rotation(self.h, key, c0.ptr, c1.ptr, k, c0.n_limbs, c0.batch,
                                                                 o0.ptr, o1.ptr, ws, wsb, stream))
        return o0, o1

    def ext_limbs(self, sizeQl):
        """context limbs of the extended basis Q_l u P"""
        return np.concatenate([np.arange(sizeQl), np.arange(self.sizeQ, self.sizeQ + self.sizeP)]).astype(np.uint32)

    def KeySwitchExt(self, c, stream=None):  # keyswitch-hybrid.cpp:217-243, one element
        out = self.ctx.empty(c.batch, c.n_limbs + self.sizeP, self.ext_limbs(c.n_limbs))
        self.ctx.lib.check(self.ctx.lib.L.fhe_ks_ext(self.h, c.ptr, c.n_limbs, c.batch, out.ptr, stream))
        return out

    def EvalFastRotationExt(self, key, c0, c1, k, add_first, stream=None):  # ckksrns-leveledshe.cpp:534-582 (digits in ws)
        ws, wsb = self.workspace(c0.n_limbs, c0.batch)
        idx = self.ext_limbs(c0.n_limbs)
        o0, o1 = (self.ctx.empty(c0.batch, len(idx), idx) for _ in range(2))
        self.ctx.lib.check(self.ctx.lib.L.fhe_eval_fast_rotation_ext(self.h, key, c0.ptr, c1.ptr, k, 1 if add_first else 0,
                                                                     c0.n_limbs, c0.batch, o0.ptr, o1.ptr, ws, wsb, stream))
        return o0, o1

    def KeySwitchDown(self, x0, x1, sizeQl, stream=None):  # keyswitch-hybrid.cpp:245-278
        ws, wsb = self.workspace(sizeQl, x0.batch)
        o0, o1 = self.ctx.empty(x0.batch, sizeQl), self.ctx.empty(x0.batch, sizeQl)
        self.ctx.lib.check(self.ctx.lib.L.fhe_ks_down(self.h, x0.ptr, x1.ptr, sizeQl, x0.batch, o0.ptr, o1.ptr, ws, wsb, stream))
        return o0, o1

    def BsgsTransform(self, c0, c1, in_rot, out_rot, diag, ws=None, out=None, stream=None):
        """fhe_ckks_bsgs_transform: in_rot / out_rot = lists of (automorphism index, key handle) or None for "no rotation";
        diag[i][j] = device pointer of the plaintext rows of outer step i, inner rotation j (None = absent).
        ws = (pointer, bytes) of a caller-owned workspace (needed for graph capture), else allocated per call."""
        L = self.ctx.lib.L
        nIn, nOut = len(in_rot), len(out_rot)

        def split(rots):
            ks = (C.c_uint32 * len(rots))(*[0 if r is None else int(r[0]) for r in rots])
            hs = (vp * len(rots))(*[None if r is None else r[1] for r in rots])
            return ks, hs
        inK, inH = split(in_rot)
        outK, outH = split(out_rot)
        dp = (vp * (nIn * nOut))(*[diag[i][j] for i in range(nOut) for j in range(nIn)])
        own = ws is None
        if own:
            wsb = L.fhe_ckks_bsgs_workspace_bytes(self.h, c0.n_limbs, c0.batch, nIn, nOut)
            ws = (self.ctx.malloc(wsb), wsb)
        o0, o1 = out if out is not None else (c0.like(), c0.like())
        self.ctx.lib.check(L.fhe_ckks_bsgs_transform(self.h, c0.ptr, c1.ptr, c0.n_limbs, c0.batch, nIn, inK, inH, nOut, outK,
                                                     outH, dp, o0.ptr, o1.ptr, ws[0], ws[1], stream))
        if own:
            self.ctx.sync(stream)
            self.ctx.free(ws[0])
        return o0, o1

    def EvalLinearTransform(self, A, c0, c1, bStep, rot_keys, stream=None):
        """FHECKKSRNS::EvalLinearTransform (ckksrns-fhe.cpp:1832-1882).  A = device pointers of the `slots` encoded diagonals
        (EvalLinearTransformPrecompute), rot_keys[index] = (automorphism index, key handle) for the baby steps 1..bStep-1
        and the giant steps bStep*j."""
        slots = len(A)
        gStep = -(-slots // bStep)
        in_rot = [None] + [rot_keys[i] for i in range(1, bStep)]
        out_rot = [None] + [rot_keys[bStep * j] for j in range(1, gStep)]
        diag = [[A[bStep * j + i] if bStep * j + i < slots else None for i in range(bStep)] for j in range(gStep)]
        return self.BsgsTransform(c0, c1, in_rot, out_rot, diag, stream=stream)

    def EvalAutomorphism(self, key, c0, c1, k, stream=None):  # base-leveledshe.cpp:381-422
        ws, wsb = self.workspace(c0.n_limbs, c0.batch)
        o0, o1 = c0.like(), c0.like()
        self.ctx.lib.check(self.ctx.lib.L.fhe_eval_automorphism(self.h, key, c0.ptr, c1.ptr, k, c0.n_limbs, c0.batch,
                                                                o0.ptr, o1.ptr, ws, wsb, stream))
        return o0, o1

    def ApproxModDown(self, x, sizeQl, t=0, stream=None):  # dcrtpoly-impl.h:966-1005 (t > 0: the BGV form)
        ws, wsb = self.workspace(sizeQl, x.batch)
        out = self.ctx.empty(x.batch, sizeQl)
        L = self.ctx.lib.L
        if t:
            self.ctx.lib.check(L.fhe_approx_mod_down_bgv(self.h, x.ptr, sizeQl, t, x.batch, out.ptr, ws, wsb, stream))
        else:
            self.ctx.lib.check(L.fhe_approx_mod_down(self.h, x.ptr, sizeQl, x.batch, out.ptr, ws, wsb, stream))
        return out


def rescale(ctx, x, stream=None):
    """DCRTPoly::DropLastElementAndScale on a Tower over context limbs [0, sizeQl) (dcrtpoly-impl.h:693-712)."""
    sizeQl = x.n_limbs
    need = ctx.lib.L.fhe_rescale_workspace_bytes(ctx.h, sizeQl, x.batch)
    ws = ctx.malloc(need)
    out = ctx.empty(x.batch, sizeQl - 1)
    ctx.lib.check(ctx.lib.L.fhe_rescale(ctx.h, x.ptr, sizeQl, x.batch, out.ptr, ws, need, stream))
    ctx.sync(stream)
    ctx.free(ws)
    return out


def rescale_limbs(ctx, x, scale_tab, inv_tab, stream=None):
    """DropLastElementAndScale on a Tower over ANY limbs of the context (x.limb_idx; its last limb is dropped) with the caller's tables
    QlQlInvModqlDivqlModq / qlInvModq (dcrtpoly-impl.h:693-694), the entry the DCRTPoly backend calls."""
    sizeQl = x.n_limbs
    need = ctx.lib.L.fhe_rescale_workspace_bytes(ctx.h, sizeQl, x.batch)
    ws = ctx.malloc(need)
    kept = None if x.limb_idx is None else x.limb_idx[:sizeQl - 1]
    out = ctx.empty(x.batch, sizeQl - 1, kept)
    a = np.ascontiguousarray(scale_tab, dtype=np.uint64)
    b = np.ascontiguousarray(inv_tab, dtype=np.uint64)
    ctx.lib.check(ctx.lib.L.fhe_rescale_limbs(ctx.h, x.ptr, x._li(), sizeQl, a.ctypes.data_as(u64p), b.ctypes.data_as(u64p), x.batch,
                                              out.ptr, ws, need, stream))
    ctx.sync(stream)
    ctx.free(ws)
    return out


def rescale_limbs_pair(ctx, x0, x1, scale_tab, inv_tab, stream=None):
    """the two elements of a ciphertext (two Towers of batch 1, allocated on their own) through fhe_rescale_limbs_pair"""
    sizeQl = x0.n_limbs
    need = ctx.lib.L.fhe_rescale_workspace_bytes(ctx.h, sizeQl, 2)
    ws = ctx.malloc(need)
    kept = None if x0.limb_idx is None else x0.limb_idx[:sizeQl - 1]
    o0, o1 = ctx.empty(1, sizeQl - 1, kept), ctx.empty(1, sizeQl - 1, kept)
    a = np.ascontiguousarray(scale_tab, dtype=np.uint64)
    b = np.ascontiguousarray(inv_tab, dtype=np.uint64)
    ctx.lib.check(ctx.lib.L.fhe_rescale_limbs_pair(ctx.h, x0.ptr, x1.ptr, x0._li(), sizeQl, a.ctypes.data_as(u64p), b.ctypes.data_as(u64p),
                                                   o0.ptr, o1.ptr, ws, need, stream))
    ctx.sync(stream)
    ctx.free(ws)
    return o0, o1


def lincomb(ctx, towers, consts, accumulate_into=None, stream=None):
    """fhe_lincomb: sum_i consts[i] (.) towers[i] per limb (pke's weighted sum, ckksrns-advancedshe.cpp:97-136) in one launch per 16 terms;
    consts[i][r] = the constant of term i on limb r (reduced).  accumulate_into: a Tower the sum is added to (returned), else a new one."""
    t0 = towers[0]
    out = accumulate_into if accumulate_into is not None else ctx.empty(t0.batch, t0.n_limbs, t0.limb_idx, t0.fmt)
    ptrs = (vp * len(towers))(*[t.ptr.value if hasattr(t.ptr, "value") else t.ptr for t in towers])
    k = np.ascontiguousarray(consts, dtype=np.uint64).reshape(len(towers), t0.n_limbs)
    ctx.lib.check(ctx.lib.L.fhe_lincomb(ctx.h, out.ptr, ptrs, k.ctypes.data_as(u64p), len(towers), t0._li(), t0.n_limbs, t0.batch,
                                         1 if accumulate_into is not None else 0, stream))
    ctx.sync(stream)
    return out


def elem_pair(ctx, kind, a0, a1, b0=None, b1=None, consts=None, in_place=False, stream=None):
    """fhe_add_pair / fhe_sub_pair / fhe_mul_const_pair on Towers of batch 1 allocated on their own"""
    n = a0.n_limbs
    o0, o1 = (a0, a1) if in_place else (ctx.empty(1, n, a0.limb_idx, a0.fmt), ctx.empty(1, n, a0.limb_idx, a0.fmt))
    if kind == "mul_const":
        k = np.ascontiguousarray(consts, dtype=np.uint64)
        ctx.lib.check(ctx.lib.L.fhe_mul_const_pair(ctx.h, o0.ptr, o1.ptr, a0.ptr, a1.ptr, k.ctypes.data_as(u64p), a0._li(), n, stream))
    else:
        f = ctx.lib.L.fhe_add_pair if kind == "add" else ctx.lib.L.fhe_sub_pair
        ctx.lib.check(f(ctx.h, o0.ptr, o1.ptr, a0.ptr, a1.ptr, b0.ptr, b1.ptr, a0._li(), n, stream))
    ctx.sync(stream)
    return o0, o1


def mod_reduce(ctx, x, t, stream=None):
    """DCRTPoly::ModReduce (BGV modulus switch by the last limb, dcrtpoly-impl.h:736-755) on a Tower over limbs [0, sizeQl)."""
    sizeQl = x.n_limbs
    need = ctx.lib.L.fhe_rescale_workspace_bytes(ctx.h, sizeQl, x.batch)
    ws = ctx.malloc(need)
    out = ctx.empty(x.batch, sizeQl - 1, None, x.fmt)
    ctx.lib.check(ctx.lib.L.fhe_mod_reduce(ctx.h, x.ptr, sizeQl, t, 1 if x.fmt == EVALUATION else 0, x.batch, out.ptr, ws,
                                           need, stream))
    ctx.sync(stream)
    ctx.free(ws)
    return out


def scale_and_round_native(ctx, x, t, tab_modt, frac, tab_bmodt=None, bfrac=None, stream=None):
    """DCRTPoly::ScaleAndRound -> NativePoly mod t (BFV decryption, dcrtpoly-impl.h:1190-1467); returns uint64 [batch][N]"""
    ta = np.ascontiguousarray(tab_modt, dtype=np.uint64)
    fr = np.ascontiguousarray(frac, dtype=np.float64)
    tb = np.ascontiguousarray(tab_bmodt, dtype=np.uint64) if tab_bmodt is not None else None
    bf = np.ascontiguousarray(bfrac, dtype=np.float64) if bfrac is not None else None
    out = ctx.malloc(x.batch * ctx.N * 8)
    li = x.limb_idx.ctypes.data_as(u32p) if x.limb_idx is not None else None
    f64p = C.POINTER(C.c_double)
    ctx.lib.check(ctx.lib.L.fhe_scale_and_round_native(ctx.h, x.ptr, li, x.n_limbs, t, ta.ctypes.data_as(u64p),
                                                       tb.ctypes.data_as(u64p) if tb is not None else None,
                                                       fr.ctypes.data_as(f64p), bf.ctypes.data_as(f64p) if bf is not None else None,
                                                       x.batch, out, stream))
    host = np.empty((x.batch, ctx.N), np.uint64)
    ctx.lib.check(ctx.lib.L.fhe_memcpy_d2h(ctx.h, host.ctypes.data_as(vp), out, host.nbytes, stream))
    ctx.sync(stream)
    ctx.free(out)
    return host


def scale_and_round_behz_decrypt(ctx, x, tgamma, tgamma_qhat_modq, neg_invq_mod_tgamma, stream=None):
    """DCRTPoly::ScaleAndRound, BEHZ decryption overload (dcrtpoly-impl.h:1631-1671); returns uint64 [batch][N]"""
    ta = np.ascontiguousarray(tgamma_qhat_modq, dtype=np.uint64)
    tb = np.ascontiguousarray(neg_invq_mod_tgamma, dtype=np.uint64)
    out = ctx.malloc(x.batch * ctx.N * 8)
    li = x.limb_idx.ctypes.data_as(u32p) if x.limb_idx is not None else None
    ctx.lib.check(ctx.lib.L.fhe_scale_and_round_behz_decrypt(ctx.h, x.ptr, li, x.n_limbs, tgamma, ta.ctypes.data_as(u64p),
                                                             tb.ctypes.data_as(u64p), x.batch, out, stream))
    host = np.empty((x.batch, ctx.N), np.uint64)
    ctx.lib.check(ctx.lib.L.fhe_memcpy_d2h(ctx.h, host.ctypes.data_as(vp), out, host.nbytes, stream))
    ctx.sync(stream)
    ctx.free(out)
    return host


class ScaleAndRoundPlan:
    """DCRTPoly::ScaleAndRound / ApproxScaleAndRound with the reference's tables (dcrtpoly-impl.h:1470-1628)."""

    def __init__(self, ctx, sizeI, out_idx, tab, frac=None):
        self.ctx, self.sizeI = ctx, sizeI
        self.out_idx = np.ascontiguousarray(np.asarray(out_idx, dtype=np.uint32))
        tab = np.ascontiguousarray(tab, dtype=np.uint64)
        assert tab.shape == (len(self.out_idx), sizeI + 1)
        fp = None
        if frac is not None:
            frac = np.ascontiguousarray(frac, dtype=np.float64)
            fp = frac.ctypes.data_as(C.POINTER(C.c_double))
        h = vp()
        ctx.lib.check(ctx.lib.L.fhe_sr_plan_create(ctx.h, sizeI, self.out_idx.ctypes.data_as(u32p), len(self.out_idx),
                                                   tab.ctypes.data_as(u64p), fp, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.ctx.lib.L.fhe_sr_plan_destroy(self.h)
            self.h = None

    def run(self, x, output_first, stream=None):
        out = self.ctx.empty(x.batch, len(self.out_idx), self.out_idx, COEFFICIENT)
        self.ctx.lib.check(self.ctx.lib.L.fhe_scale_and_round(self.h, x.ptr, 1 if output_first else 0, out.ptr, x.batch, stream))
        return out


def scale_and_round_p_over_q(ctx, x, limb_idx, stream=None):
    """DCRTPoly::ScaleAndRoundPOverQ (dcrtpoly-impl.h:1674-1689): x over limb_idx (sizeQ+1 limbs) -> sizeQ limbs"""
    li = np.ascontiguousarray(np.asarray(limb_idx, dtype=np.uint32))
    sizeQ = len(li) - 1
    out = ctx.empty(x.batch, sizeQ, li[:sizeQ], COEFFICIENT)
    ctx.lib.check(ctx.lib.L.fhe_scale_and_round_p_over_q(ctx.h, x.ptr, li.ctypes.data_as(u32p), sizeQ, out.ptr, x.batch, stream))
    return out


class Behz:
    """BEHZ base conversions over a context that holds the Q limbs and the Bsk limbs."""

    def __init__(self, ctx, q_idx, bsk_idx, t):
        self.ctx = ctx
        self.q_idx = np.ascontiguousarray(np.asarray(q_idx, dtype=np.uint32))
        self.bsk_idx = np.ascontiguousarray(np.asarray(bsk_idx, dtype=np.uint32))
        self.numQ, self.numBsk = len(self.q_idx), len(self.bsk_idx)
        h = vp()
        ctx.lib.check(ctx.lib.L.fhe_behz_create(ctx.h, self.q_idx.ctypes.data_as(u32p), self.numQ,
                                                self.bsk_idx.ctypes.data_as(u32p), t, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.ctx.lib.L.fhe_behz_destroy(self.h)
            self.h = None

    def FastBaseConvqToBskMontgomery(self, xq_host, eval_format, stream=None):
        """xq_host: uint64 [batch][numQ][N]; returns the extended tower [batch][numQ+numBsk][N] (EVALUATION)"""
        xq_host = np.asarray(xq_host, dtype=np.uint64)
        B, N = xq_host.shape[0], self.ctx.N
        full = np.zeros((B, self.numQ + self.numBsk, N), np.uint64)
        full[:, :self.numQ] = xq_host
        t = self.ctx.tower(full, limb_idx=np.concatenate([self.q_idx, self.bsk_idx]))
        wsb = self.ctx.lib.L.fhe_behz_workspace_bytes(self.h, B)
        ws = self.ctx.malloc(wsb)
        self.ctx.lib.check(self.ctx.lib.L.fhe_behz_q_to_bsk(self.h, t.ptr, 1 if eval_format else 0, B, ws, wsb, stream))
        self.ctx.sync(stream)
        self.ctx.free(ws)
        t.fmt = EVALUATION
        return t

    def FastRNSFloorq(self, t, stream=None):
        self.ctx.lib.check(self.ctx.lib.L.fhe_behz_floorq(self.h, t.ptr, t.batch, stream))
        return t

    def FastBaseConvSK(self, t, stream=None):
        out = self.ctx.empty(t.batch, self.numQ, self.q_idx, COEFFICIENT)
        self.ctx.lib.check(self.ctx.lib.L.fhe_behz_conv_sk(self.h, t.ptr, out.ptr, t.batch, stream))
        return out

    def EvalMult(self, ks_plan, a0, a1, b0, b1, stream=None):
        """LeveledSHEBase::EvalMult(ct, ct, key) on BFV/BEHZ ciphertexts: EvalMultNoRelin + relinearisation with the
        key-switch plan's evaluation key; returns (c0, c1), EVALUATION"""
        B = a0.batch
        c0, c1 = (self.ctx.empty(B, self.numQ, self.q_idx, EVALUATION) for _ in range(2))
        L = self.ctx.lib.L
        wsb = L.fhe_bfv_eval_mult_relin_workspace_bytes(self.h, ks_plan.h, B)
        ws = self.ctx.malloc(wsb)
        try:
            self.ctx.lib.check(L.fhe_bfv_eval_mult_relin_behz(self.h, ks_plan.h, ks_plan.key, a0.ptr, a1.ptr, b0.ptr, b1.ptr,
                                                              c0.ptr, c1.ptr, B, ws, wsb, stream))
            self.ctx.sync(stream)
        finally:
            self.ctx.free(ws)
        return c0, c1

    def EvalMultNoRelin(self, a0, a1, b0, b1, out_eval=False, stream=None):
        """LeveledSHEBFVRNS::EvalMult (BEHZ) on device towers [batch][numQ][N] (EVALUATION); returns (d0, d1, d2)"""
        B = a0.batch
        fmt = EVALUATION if out_eval else COEFFICIENT
        d = [self.ctx.empty(B, self.numQ, self.q_idx, fmt) for _ in range(3)]
        L = self.ctx.lib.L
        wsb = L.fhe_bfv_eval_mult_behz_workspace_bytes(self.h, B)
        ws = self.ctx.malloc(wsb)
        try:
            self.ctx.lib.check(L.fhe_bfv_eval_mult_behz(self.h, a0.ptr, a1.ptr, b0.ptr, b1.ptr, d[0].ptr, d[1].ptr,
                                                        d[2].ptr, 1 if out_eval else 0, B, ws, wsb, stream))
            self.ctx.sync(stream)
        finally:
            self.ctx.free(ws)
        return d
