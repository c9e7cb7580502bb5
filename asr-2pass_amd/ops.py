"""ctypes binding of the operator-level C ABI (include/pfhip_ops.h) over torch CUDA tensors.

torch is plumbing only here: it owns the device buffers and the stream; every op below runs the same
hand-written gfx950 kernel that pfhip_offline_forward launches.  No fallbacks: a missing library or a
non-CUDA tensor raises.
"""
from __future__ import annotations

import ctypes

import torch

from . import load_lib, PfhipError

_vp, _ci, _cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
_bound = False


def _lib():
    global _bound
    lib = load_lib()
    if not _bound:
        lib.pfhip_op_gemm_f32.argtypes = [_vp, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _ci, _vp, _ci, _ci, _ci, _ci, _ci, _ci, _vp]
        lib.pfhip_op_gemm_f32_kind.argtypes = [_vp, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _ci, _vp, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _vp]
        lib.pfhip_op_fused_ln_gemm.argtypes = [_vp, _ci, _ci, _vp, _vp, _cf, _vp, _ci, _vp, _ci, _vp, _vp, _ci, _vp, _ci, _vp, _ci, _vp,
                                               _ci, _ci, _ci, _ci, _vp]
        lib.pfhip_op_fused_gemv_1trip.argtypes = [_vp, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _cf, _vp, _ci, _vp, _ci, _vp, _ci, _ci, _ci, _ci, _vp]
        lib.pfhip_op_window_attention.argtypes = [_vp, _ci, _vp, _ci, _vp, _ci, _vp, _ci, _ci, _ci, _ci, _cf, _vp]
        lib.pfhip_op_fused_att_out.argtypes = [_vp, _ci, _vp, _ci, _vp, _ci, _ci, _ci, _ci, _cf, _vp, _ci, _vp, _ci, _vp, _vp, _ci, _vp, _ci, _vp,
                                               _ci, _vp]
        lib.pfhip_op_layernorm.argtypes = [_vp, _ci, _vp, _ci, _vp, _vp, _ci, _ci, _ci, _cf, _vp]
        lib.pfhip_op_fsmn.argtypes = [_vp, _ci, _vp, _vp, _ci, _vp, _ci, _vp, _vp, _ci, _ci, _ci, _vp]
        lib.pfhip_op_attention.argtypes = [_vp, _ci, _vp, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _cf, _vp]
        lib.pfhip_op_attention_hd.argtypes = [_vp, _ci, _vp, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _cf, _ci, _vp]
        lib.pfhip_op_cif.argtypes = [_vp, _ci, _vp, _vp, _vp, _ci, _ci, _cf, _cf, _vp, _vp, _vp, _vp]
        lib.pfhip_op_logsoftmax_argmax.argtypes = [_vp, _ci, _ci, _ci, _vp, _vp, _vp]
        _bound = True
    return lib


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise PfhipError("pfhip ops need CUDA (HIP) tensors; there is no CPU path")
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ck(rc, what):
    if rc != 0:
        raise PfhipError(f"{what}: hip error {rc}")


def round_up(v, m):
    return (v + m - 1) // m * m


def best_w_scale(max_abs):
    lib = _lib()
    lib.pfhip_op_best_w_scale.restype = ctypes.c_float
    lib.pfhip_op_best_w_scale.argtypes = [ctypes.c_float]
    return float(lib.pfhip_op_best_w_scale(float(max_abs)))


def gemm_f32(A, W, bias=None, R1=None, R2=None, relu=False, M=None, N=None, guard=True, out=None, kind=0, w_scale=None):
    """C[M,N] = A[M,K] @ W[N,K]^T (+bias +R1 +R2, ReLU).  A must have ceil(M/128)*128 rows allocated
    and W ceil(N/128)*128 rows; K % 32 == 0.  w_scale: the power-of-two weight scale of the fp16 two-plane kernels
    ("auto" = best_w_scale(max |W|), what the model does per weight matrix at load; None = 1)."""
    K = A.shape[1]
    M = A.shape[0] if M is None else M
    N = W.shape[0] if N is None else N
    if out is None:
        out = torch.empty((round_up(M, 128), round_up(N, 128)), dtype=torch.float32, device=A.device)
    if w_scale is not None:
        if w_scale == "auto":
            w_scale = best_w_scale(float(W.abs().max()))
        lib = _lib()
        lib.pfhip_op_gemm_f32_scaled.argtypes = [_vp, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _ci, _vp, _ci, _ci, _ci, _ci, _ci, _ci, _ci,
                                                 ctypes.c_float, _vp]
        _ck(lib.pfhip_op_gemm_f32_scaled(_p(A), A.stride(0), _p(W), W.stride(0), _p(out), out.stride(0), _p(bias),
                                         _p(R1), R1.stride(0) if R1 is not None else 0, _p(R2),
                                         R2.stride(0) if R2 is not None else 0, M, N, K, 1 if relu else 0,
                                         1 if guard else 0, kind, float(w_scale), _stream()), "gemm")
        return out
    _ck(_lib().pfhip_op_gemm_f32_kind(_p(A), A.stride(0), _p(W), W.stride(0), _p(out), out.stride(0), _p(bias),
                                      _p(R1), R1.stride(0) if R1 is not None else 0, _p(R2),
                                      R2.stride(0) if R2 is not None else 0, M, N, K, 1 if relu else 0,
                                      1 if guard else 0, kind, _stream()), "gemm")
    return out


def fused_ln_gemm(X, W, M, N, g=None, b=None, D=None, bias=None, R1=None, R2=None, fsmn_v=None, fsmn_w=None, relu=False, out=None,
                  eps=1e-12):
    """One streaming window: LN (if g) -> GEMM (+bias +R1 +R2 +FSMN memory of fsmn_v, ReLU) in one launch (M <= 32)."""
    K = W.shape[1]
    if out is None:
        out = torch.zeros((32, round_up(N, 128)), dtype=torch.float32, device=X.device)
    _ck(_lib().pfhip_op_fused_ln_gemm(_p(X), X.stride(0), K if D is None else D, _p(g), _p(b), eps, _p(W), W.stride(0), _p(out),
                                      out.stride(0), _p(bias), _p(R1), R1.stride(0) if R1 is not None else 0, _p(R2),
                                      R2.stride(0) if R2 is not None else 0, _p(fsmn_v), fsmn_v.stride(0) if fsmn_v is not None else 0,
                                      _p(fsmn_w), M, N, K, 1 if relu else 0, _stream()), "fused_ln_gemm")
    return out


def fold_layernorm(W, bias, g, b):
    """(W', b', colsum) for fused_gemv_1trip's algebraic LayerNorm: W' = W * g, b' = bias + W @ b, colsum = W'.sum(1) — in
    double, rounded once, as the library does at load (pfhip.cpp)."""
    Wd = W.double()
    Wf = (Wd * g.double()[None, :]).float()
    bf = ((bias.double() if bias is not None else 0.0) + Wd @ b.double()).float()
    return Wf, bf, Wf.double().sum(1).float()


def fused_gemv_1trip(X, W, M, N, bias=None, ln_colsum=None, R1=None, fsmn_v=None, fsmn_w=None, relu=False, out=None, eps=1e-12):
    """One streaming window (M <= 20): (LN ->) GEMM (+bias +R1 +FSMN memory, ReLU), operands requested in one trip; with
    ln_colsum, W / bias are the folded ones of fold_layernorm."""
    K = W.shape[1]
    if out is None:
        out = torch.zeros((32, round_up(N, 128)), dtype=torch.float32, device=X.device)
    _ck(_lib().pfhip_op_fused_gemv_1trip(_p(X), X.stride(0), _p(W), W.stride(0), _p(out), out.stride(0), _p(bias), _p(ln_colsum), eps,
                                         _p(R1), R1.stride(0) if R1 is not None else 0, _p(fsmn_v),
                                         fsmn_v.stride(0) if fsmn_v is not None else 0, _p(fsmn_w), M, N, K, 1 if relu else 0, _stream()),
        "fused_gemv_1trip")
    return out


def layernorm(x, g, b, D=None, Dout=None, eps=1e-12):
    M = x.shape[0]
    D = x.shape[1] if D is None else D
    Dout = D if Dout is None else Dout
    y = torch.empty((M, Dout), dtype=torch.float32, device=x.device)
    _ck(_lib().pfhip_op_layernorm(_p(x), x.stride(0), _p(y), y.stride(0), _p(g), _p(b), M, D, Dout, eps, _stream()), "layernorm")
    return y


def fsmn(v, w, off, length, res=None):
    out = torch.zeros((v.shape[0], v.shape[1]), dtype=torch.float32, device=v.device)
    B = off.numel()
    _ck(_lib().pfhip_op_fsmn(_p(v), v.stride(0), _p(w), _p(res), res.stride(0) if res is not None else 0, _p(out),
                             out.stride(0), _p(off), _p(length), B, int(length.max().item()), v.shape[1], _stream()), "fsmn")
    return out


def attention(Q, K, V, q_off, q_len, kv_off, kv_len, n_head, scale, head_dim=128):
    O = torch.zeros((Q.shape[0], n_head * head_dim), dtype=torch.float32, device=Q.device)
    B = q_off.numel()
    _ck(_lib().pfhip_op_attention_hd(_p(Q), Q.stride(0), _p(K), K.stride(0), _p(V), V.stride(0), _p(O), O.stride(0),
                                     _p(q_off), _p(q_len), _p(kv_off), _p(kv_len), B, n_head, int(q_len.max().item()),
                                     scale, head_dim, _stream()), "attention")
    return O


def attention_planes(Q, K, V, q_off, q_len, kv_off, kv_len, n_head, scale):
    """attention (d_k = 128) with the context as fp16 plane images: returns (hi, lo, rows) as split_planes does."""
    lib = _lib()
    rows = round_up(Q.shape[0], 128)
    lib.pfhip_op_plane_image_bytes.restype = ctypes.c_size_t
    lib.pfhip_op_plane_image_bytes.argtypes = [_ci, _ci]
    nb = int(lib.pfhip_op_plane_image_bytes(rows, n_head * 128))
    hi = torch.zeros(nb, dtype=torch.uint8, device=Q.device)
    lo = torch.zeros(nb, dtype=torch.uint8, device=Q.device)
    lib.pfhip_op_attention_planes.argtypes = [_vp, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _ci, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci,
                                             ctypes.c_float, _vp]
    _ck(lib.pfhip_op_attention_planes(_p(Q), Q.stride(0), _p(K), K.stride(0), _p(V), V.stride(0), _p(hi), _p(lo), rows, _p(q_off),
                                      _p(q_len), _p(kv_off), _p(kv_len), q_off.numel(), n_head, int(q_len.max().item()), Q.shape[0],
                                      float(scale), _stream()), "attention_planes")
    return hi, lo, rows


def window_attention(Q, K, V, Lq, Lk, n_head, scale):
    """One streaming window: softmax(scale Q_h K_h^T) V_h for rows [0, Lq) x [0, Lk), d_k = 128."""
    O = torch.zeros((Q.shape[0], n_head * 128), dtype=torch.float32, device=Q.device)
    _ck(_lib().pfhip_op_window_attention(_p(Q), Q.stride(0), _p(K), K.stride(0), _p(V), V.stride(0), _p(O), O.stride(0), Lq, Lk, n_head,
                                         scale, _stream()), "window_attention")
    return O


def fused_att_out(Q, K, V, Lq, Lk, n_head, scale, W, bias=None, R1=None, fsmn_v=None, fsmn_w=None, out=None):
    """One streaming window: attention (Lq x Lk, 4 heads of 128) and the projection of its context by W [N, 512] in one launch."""
    N = W.shape[0]
    if out is None:
        out = torch.zeros((32, N), dtype=torch.float32, device=Q.device)
    _ck(_lib().pfhip_op_fused_att_out(_p(Q), Q.stride(0), _p(K), K.stride(0), _p(V), V.stride(0), Lq, Lk, n_head, scale, _p(W), W.stride(0),
                                      _p(out), out.stride(0), _p(bias), _p(R1), R1.stride(0) if R1 is not None else 0, _p(fsmn_v),
                                      fsmn_v.stride(0) if fsmn_v is not None else 0, _p(fsmn_w), N, _stream()), "fused_att_out")
    return out


def cif(hidden, alphas, row_off, length, threshold, tail):
    B = row_off.numel()
    D = hidden.shape[1]
    stage = torch.zeros((hidden.shape[0] + B, D), dtype=torch.float32, device=hidden.device)
    n_fires = torch.zeros(B, dtype=torch.int32, device=hidden.device)
    token_num = torch.zeros(B, dtype=torch.int32, device=hidden.device)
    _ck(_lib().pfhip_op_cif(_p(hidden), hidden.stride(0), _p(alphas), _p(row_off), _p(length), B, D, threshold, tail,
                            _p(stage), _p(n_fires), _p(token_num), _stream()), "cif")
    return stage, n_fires, token_num


def logsoftmax_argmax(logits, V=None, want_logp=True):
    ML = logits.shape[0]
    V = logits.shape[1] if V is None else V
    logp = torch.empty((ML, V), dtype=torch.float32, device=logits.device) if want_logp else None
    ids = torch.empty(ML, dtype=torch.int32, device=logits.device)
    _ck(_lib().pfhip_op_logsoftmax_argmax(_p(logits), logits.stride(0), ML, V, _p(logp), _p(ids), _stream()), "logsoftmax")
    return logp, ids


# ---- pre-split operands (csrc/gemm_p3.hip) ------------------------------------------------------------------------------------
def split_planes(X, rows=None, scale=1.0, rows_valid=None):
    """fp32 [R, K] (device) -> (hi, lo) plane images (uint8 tensors of plane_image_bytes each), rows padded to a multiple of 128."""
    lib = _lib()
    R, K = X.shape
    rows_valid = R if rows_valid is None else rows_valid
    rows = round_up(R if rows is None else rows, 128)
    lib.pfhip_op_plane_image_bytes.restype = ctypes.c_size_t
    lib.pfhip_op_plane_image_bytes.argtypes = [_ci, _ci]
    nb = int(lib.pfhip_op_plane_image_bytes(rows, K))
    hi = torch.empty(nb, dtype=torch.uint8, device=X.device)
    lo = torch.empty(nb, dtype=torch.uint8, device=X.device)
    lib.pfhip_op_split_planes.argtypes = [_vp, _ci, _ci, _ci, _ci, ctypes.c_float, _vp, _vp, _vp]
    _ck(lib.pfhip_op_split_planes(_p(X), X.stride(0), rows_valid, rows, K, float(scale), _p(hi), _p(lo), _stream()), "split_planes")
    return hi, lo, rows


def planes_to_float(hi, lo, rows, K):
    """Inverse of the image layout (host side, for tests): -> float64 [rows, K] = hi + lo."""
    import numpy as np
    out = np.zeros((rows, K), np.float64)
    for img in (hi, lo):
        a = img.cpu().numpy().view(np.float16).reshape(K // 16, rows, 2, 8).astype(np.float64)
        swap = ((np.arange(rows) >> 3) & 1).astype(bool)
        a[:, swap] = a[:, swap][:, :, ::-1]
        out += a.transpose(1, 0, 2, 3).reshape(rows, K)
    return out


def gemm_p3(A_img, W_img, M, N, K, w_scale=1.0, bias=None, R1=None, relu=False, want_c=True, want_planes=False, ln_stats=None,
            ln_tiles=0, ln_colsum=None, stats_out=None, out=None, out_planes=None, tile_rows=0):
    """A_img / W_img: (hi, lo, rows) from split_planes.  Returns (C or None, (hi, lo, rows) of C or None).  out / out_planes: reuse
    buffers of an earlier call (timing loops)."""
    lib = _lib()
    ah, al, ra = A_img
    wh, wl, rw = W_img
    Mp = round_up(M, 128)
    C = (out if out is not None else torch.empty((Mp, N), dtype=torch.float32, device=ah.device)) if want_c else None
    P = None
    if want_planes and out_planes is not None:
        P = out_planes
    elif want_planes:
        nb = int(lib.pfhip_op_plane_image_bytes(Mp, N))
        P = (torch.zeros(nb, dtype=torch.uint8, device=ah.device), torch.zeros(nb, dtype=torch.uint8, device=ah.device), Mp)
    lib.pfhip_op_gemm_p3.argtypes = [_vp, _vp, _ci, _vp, _vp, _ci, ctypes.c_float, _vp, _ci, _vp, _vp, _ci, _vp, _vp, _ci, _ci, _ci, _ci, _ci,
                                     _vp, _ci, _vp, _vp, _ci, _vp]
    _ck(lib.pfhip_op_gemm_p3(_p(ah), _p(al), ra, _p(wh), _p(wl), rw, float(w_scale), _p(C), N if want_c else 0, _p(P[0]) if P else None,
                             _p(P[1]) if P else None, Mp, _p(bias), _p(R1), R1.stride(0) if R1 is not None else 0, M, N, K,
                             1 if relu else 0, _p(ln_stats), ln_tiles, _p(ln_colsum), _p(stats_out), tile_rows, _stream()), "gemm_p3")
    return C, P


# ---- K | V as row-major planes (csrc/attention_p3.hip) ------------------------------------------------------------------------------
def split_rows(X, ldp=None, out=None, col=0):
    """fp32 [R, C] (device) -> row-major planes (hi, lo): float16 tensors [R, ldp], X in columns col .. col + C - 1."""
    lib = _lib()
    R, C = X.shape
    ldp = C if ldp is None else ldp
    if out is None:
        out = (torch.zeros((R, ldp), dtype=torch.float16, device=X.device), torch.zeros((R, ldp), dtype=torch.float16, device=X.device))
    hi, lo = out
    lib.pfhip_op_split_rows.argtypes = [_vp, _ci, _ci, _ci, _vp, _vp, _ci, _vp]
    _ck(lib.pfhip_op_split_rows(_p(X), X.stride(0), R, C, hi.data_ptr() + 2 * col, lo.data_ptr() + 2 * col, ldp, _stream()), "split_rows")
    return hi, lo


def gemm_p3_qkv(A_img, W_img, M, N, K, q_cols, w_scale=1.0, bias=None, ln_stats=None, ln_tiles=0, ln_colsum=None, tile_rows=0):
    """The QKV projection on plane-image operands: returns (C [Mp, q_cols] fp32, (kv_hi, kv_lo) float16 [Mp, N - q_cols])."""
    lib = _lib()
    ah, al, ra = A_img
    wh, wl, rw = W_img
    Mp = round_up(M, 128)
    C = torch.zeros((Mp, q_cols), dtype=torch.float32, device=ah.device)
    kvh = torch.zeros((Mp, N - q_cols), dtype=torch.float16, device=ah.device)
    kvl = torch.zeros((Mp, N - q_cols), dtype=torch.float16, device=ah.device)
    lib.pfhip_op_gemm_p3_qkv.argtypes = [_vp, _vp, _ci, _vp, _vp, _ci, ctypes.c_float, _vp, _ci, _vp, _vp, _ci, _ci, _vp, _ci, _ci, _ci, _vp, _ci,
                                         _vp, _ci, _vp]
    _ck(lib.pfhip_op_gemm_p3_qkv(_p(ah), _p(al), ra, _p(wh), _p(wl), rw, float(w_scale), _p(C), q_cols, _p(kvh), _p(kvl), N - q_cols, q_cols,
                                 _p(bias), M, N, K, _p(ln_stats), ln_tiles, _p(ln_colsum), tile_rows, _stream()), "gemm_p3_qkv")
    return C, (kvh, kvl)


def attention_kvplanes(Q, kv, v_col, q_off, q_len, kv_off, kv_len, n_head, scale, fsmn_w=None, mem=None, mem_accumulate=False,
                       want_planes=False):
    """attention (d_k = 128) on K | V given as row-major planes kv = (hi, lo) float16 [R, ld] (K at column 0, V at column v_col).
    Returns the context as fp32 rows, or as (hi, lo, rows) plane images with want_planes."""
    lib = _lib()
    kvh, kvl = kv
    O, P = None, None
    rows = round_up(Q.shape[0], 128)
    if want_planes:
        lib.pfhip_op_plane_image_bytes.restype = ctypes.c_size_t
        lib.pfhip_op_plane_image_bytes.argtypes = [_ci, _ci]
        nb = int(lib.pfhip_op_plane_image_bytes(rows, n_head * 128))
        P = (torch.zeros(nb, dtype=torch.uint8, device=Q.device), torch.zeros(nb, dtype=torch.uint8, device=Q.device), rows)
    else:
        O = torch.zeros((Q.shape[0], n_head * 128), dtype=torch.float32, device=Q.device)
    lib.pfhip_op_attention_kvplanes.argtypes = [_vp, _ci, _vp, _vp, _ci, _ci, _ci, _vp, _ci, _vp, _vp, _ci, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci,
                                                ctypes.c_float, _vp, _vp, _ci, _ci, _vp]
    _ck(lib.pfhip_op_attention_kvplanes(_p(Q), Q.stride(0), _p(kvh), _p(kvl), kvh.stride(0), v_col, kvh.shape[0], _p(O),
                                        O.stride(0) if O is not None else 0, _p(P[0]) if P else None, _p(P[1]) if P else None, rows,
                                        _p(q_off), _p(q_len), _p(kv_off), _p(kv_len), q_off.numel(), n_head, int(q_len.max().item()),
                                        Q.shape[0], float(scale), _p(fsmn_w), _p(mem), mem.stride(0) if mem is not None else 0,
                                        1 if mem_accumulate else 0, _stream()), "attention_kvplanes")
    return P if want_planes else O
