"""Layout algebra of the register-resident MLP chain (csrc/mlp_chain.hip), checked on the CPU with a lane-level model of
v_mfma_f32_32x32x16_bf16's operand layout (the layout the shipped kernels csrc/split_gemm.hip / split_wgrad.hip are built on
and which tests/test_gpu_split_gemm.py pins on the hardware):

    A fragment: lane l holds A[m = l & 31][k = 8 (l >> 5) + i], i = 0..7
    B fragment: lane l holds B[k = 8 (l >> 5) + i][n = l & 31]
    C / D     : lane l, register r holds D[m = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][n = l & 31]

The chain keeps one wave's 32 batch rows in the N dimension ("lane = batch row") and lets every layer's OUTPUT registers be the
next product's operand as they are: a lane's accumulator registers 8 s .. 8 s + 7 of output tile t are eight K elements of the
next product, in the permuted order k = 16 (2 t + s) + perm(h, i), perm(h, i) = (i & 3) + 8 (i >> 2) + 4 h - any enumeration of
the contraction index is valid as long as the other operand (the weights, prepared once per optimizer step) uses the same one.
This file states the index formulas the kernel and the weight-preparation kernels implement and verifies them against plain
matrix products (float64; the bf16 splitting is orthogonal to the layout and is tested on the GPU)."""
import numpy as np

LANES = np.arange(64)
L31, H = LANES & 31, LANES >> 5


def mfma(a_frag, b_frag, c):
    """a_frag, b_frag: [64, 8]; c: [64, 16] -> d [64, 16] (one 32x32x16 product-accumulate)."""
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for l in range(64):
        A[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = a_frag[l]
        B[8 * (l >> 5):8 * (l >> 5) + 8, l & 31] = b_frag[l]
    D = A @ B
    d = c.copy()
    for l in range(64):
        for r in range(16):
            d[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    return d


def row_of_reg(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def perm(h, i):
    return (i & 3) + 8 * (i >> 2) + 4 * h


def elu(z):
    return np.where(z > 0, z, np.exp(np.minimum(z, 0)) - 1)


def test_chain_forward_backward_layouts():
    rng = np.random.default_rng(0)
    D, F, A1, R = 18, 256, 5, 32                      # inputs, layer width, heads (4 mu + value), batch rows of one wave
    x = rng.normal(size=(R, D)); W1 = rng.normal(size=(F, D)) * 0.3; b1 = rng.normal(size=F) * 0.1
    W2 = rng.normal(size=(F, F)) * 0.08; b2 = rng.normal(size=F) * 0.1
    Wh = rng.normal(size=(A1, F)) * 0.1; bh = rng.normal(size=A1) * 0.1
    # ---- reference (row-major batch)
    h1 = elu(x @ W1.T + b1); z2 = h1 @ W2.T + b2; h2 = elu(z2); heads = h2 @ Wh.T + bh
    dheads = rng.normal(size=(R, A1))                 # stands for d loss / d heads
    dh2 = dheads @ Wh; dz2 = dh2 * np.where(h2 > 0, 1.0, h2 + 1.0)
    dh1 = dz2 @ W2; dz1 = dh1 * np.where(h1 > 0, 1.0, h1 + 1.0)
    dW1 = dz1.T @ x; db1 = dz1.sum(0)

    # ---- layer 1, transposed: H1^T = W1e X_e^T with the bias as column D of W1e against a ones column of x (K padded to 32)
    KP = 32
    xe = np.zeros((R, KP)); xe[:, :D] = x; xe[:, D] = 1.0
    W1e = np.zeros((F, KP)); W1e[:, :D] = W1; W1e[:, D] = b1
    acc1 = [np.zeros((64, 16)) for _ in range(8)]
    for t in range(8):
        for s in range(KP // 16):
            a = np.stack([W1e[32 * t + (l & 31), 16 * s + 8 * (l >> 5):16 * s + 8 * (l >> 5) + 8] for l in range(64)])   # natural K order
            b = np.stack([xe[l & 31, 16 * s + 8 * (l >> 5):16 * s + 8 * (l >> 5) + 8] for l in range(64)])
            acc1[t] = mfma(a, b, acc1[t])
    for t in range(8):
        for l in range(64):
            for r in range(16):
                assert abs(acc1[t][l, r] - (x @ W1.T + b1)[l & 31, 32 * t + row_of_reg(r, l >> 5)]) < 1e-12
    h1r = [elu(a) for a in acc1]                      # ELU in place: lane = batch row, registers = features

    # ---- layer 2, transposed, B operand = layer 1's registers as they are; weights in chain K order
    def w_chain_unit(W, m, c, h):                     # A-operand unit of K step c for output row m, lane half h
        return np.array([W[m, 16 * c + perm(h, i)] for i in range(8)])
    acc2 = [np.zeros((64, 16)) for _ in range(8)]
    for c in range(16):
        t, s = c >> 1, c & 1
        b = np.stack([h1r[t][l, 8 * s:8 * s + 8] for l in range(64)])
        for j in range(8):
            a = np.stack([w_chain_unit(W2, 32 * j + (l & 31), c, l >> 5) for l in range(64)])
            acc2[j] = mfma(a, b, acc2[j])
    for j in range(8):                                # bias: accumulator + b2[feature of (j, r, h)]
        for l in range(64):
            for r in range(16):
                acc2[j][l, r] += b2[32 * j + row_of_reg(r, l >> 5)]
                assert abs(acc2[j][l, r] - z2[l & 31, 32 * j + row_of_reg(r, l >> 5)]) < 1e-11
    h2r = [elu(a) for a in acc2]

    # ---- heads, transposed: rows 0..4 of a 32-row tile, the rest zero weights
    Whp = np.zeros((32, F)); Whp[:A1] = Wh
    acch = np.zeros((64, 16))
    for c in range(16):
        t, s = c >> 1, c & 1
        b = np.stack([h2r[t][l, 8 * s:8 * s + 8] for l in range(64)])
        a = np.stack([w_chain_unit(Whp, l & 31, c, l >> 5) for l in range(64)])
        acch = mfma(a, b, acch)
    for l in range(64):
        for r in range(16):
            a_idx = row_of_reg(r, l >> 5)
            want = heads[l & 31, a_idx] - bh[a_idx] if a_idx < A1 else 0.0
            assert abs(acch[l, r] - want) < 1e-11
    # lane n (h = 0) holds heads 0..3 of batch row n in registers 0..3, lane n + 32 holds head 4 in register 0
    assert row_of_reg(0, 1) == 4 and [row_of_reg(r, 0) for r in range(4)] == [0, 1, 2, 3]

    # ---- dh2^T = Wh^T dheads^T: one K step over the (padded) head index, natural K order (a fresh operand, not registers)
    dhe = np.zeros((R, 16)); dhe[:, :A1] = dheads
    acc_dz2 = []
    for j in range(8):
        a = np.stack([[Whp[8 * (l >> 5) + i, 32 * j + (l & 31)] if 8 * (l >> 5) + i < 32 else 0.0 for i in range(8)] for l in range(64)])
        b = np.stack([dhe[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] for l in range(64)])
        acc = mfma(a, b, np.zeros((64, 16)))
        acc = acc * np.where(h2r[j] > 0, 1.0, h2r[j] + 1.0)          # ELU' from the activations held in the same layout
        acc_dz2.append(acc)
        for l in range(64):
            for r in range(16):
                assert abs(acc[l, r] - dz2[l & 31, 32 * j + row_of_reg(r, l >> 5)]) < 1e-11

    # ---- dX, NOT transposed: dH1 = dZ2 W2 with the dz2 registers as the A operand (lane = batch row = M) and W2 as B [K = out][N = in]
    acc_dh1 = [np.zeros((64, 16)) for _ in range(8)]
    for c in range(16):
        t, s = c >> 1, c & 1
        a = np.stack([acc_dz2[t][l, 8 * s:8 * s + 8] for l in range(64)])
        for j in range(8):
            b = np.stack([[W2[16 * c + perm(l >> 5, i), 32 * j + (l & 31)] for i in range(8)] for l in range(64)])
            acc_dh1[j] = mfma(a, b, acc_dh1[j])
    for j in range(8):                                # now lane = input feature, registers = batch rows
        for l in range(64):
            for r in range(16):
                assert abs(acc_dh1[j][l, r] - dh1[row_of_reg(r, l >> 5), 32 * j + (l & 31)]) < 1e-11

    # ---- first layer's weight / bias gradient with K = batch rows: dz1 formed in place, x (with the ones column) as the A operand
    for J in range(8):
        G = np.zeros((64, 16))
        for ks in range(2):
            b = np.zeros((64, 8)); a = np.zeros((64, 8))
            for l in range(64):
                for i in range(8):
                    row = row_of_reg(8 * ks + i, l >> 5)      # the batch row accumulator register 8 ks + i of this lane holds
                    f = 32 * J + (l & 31)
                    b[l, i] = acc_dh1[J][l, 8 * ks + i] * (1.0 if h1[row, f] > 0 else h1[row, f] + 1.0)
                    a[l, i] = xe[row, l & 31]
            G = mfma(a, b, G)
        for l in range(64):
            for r in range(16):
                d = row_of_reg(r, l >> 5)
                f = 32 * J + (l & 31)
                want = dW1[f, d] if d < D else (db1[f] if d == D else 0.0)
                assert abs(G[l, r] - want) < 1e-10


def test_first_layer_inside_the_update_gemm_layouts():
    """csrc/split_gemm.hip, FIN > 0 (ag_split_gemm_input_loss_heads_bwd): the update's forward GEMM keeps its NATURAL orientation
    (A = activations, lane = batch row; B = W2 planes, lane = output column) but produces its A operand itself, transposed:
    per block b of 32 features  h1^T[f, row] = W1ext[f, :] . x_ext[row, :]  (weights as A fragments, the lane's input row as B
    fragments).  Registers 8 q .. 8 q + 7 of that block are then ONE A unit of the main product - chunk 2 b + q, k-half h, slot i
    <-> feature 32 b + 16 q + perm(h, i) - provided the W2 planes enumerate K the same way (split_in_prepare_kernel)."""
    rng = np.random.default_rng(1)
    D, F, R = 18, 256, 32
    x = rng.normal(size=(R, D)); W1 = rng.normal(size=(F, D)) * 0.3; b1 = rng.normal(size=F) * 0.1
    W2 = rng.normal(size=(F, F)) * 0.08
    z1 = x @ W1.T + b1
    h1 = elu(z1)
    ref = h1 @ W2.T                                    # [row, out column]

    def feat(c, h, i):                                 # K enumeration of the launch: chunk c, k-half h, slot i -> feature
        return 32 * (c >> 1) + 16 * (c & 1) + perm(h, i)
    assert sorted(feat(c, h, i) for c in range(16) for h in range(2) for i in range(8)) == list(range(F))

    # W1ext image as split_in_prepare_kernel lays it out: unit (block b, K step s, h, feature m) = W1ext[32 b + m][16 s + 8 h + i];
    # (s, h) = (1, 1) - inputs 24..31 - does not exist for D <= 23: the kernel's B fragment is all zero there
    def w1_unit(b, s, h, m):
        out = np.zeros(8)
        for i in range(8):
            d = 16 * s + 8 * h + i
            out[i] = W1[32 * b + m, d] if d < D else (b1[32 * b + m] if d == D else 0.0)
        return out

    def x_unit(row, s, h):                             # the lane's row of inputs, the all-ones column at D
        out = np.zeros(8)
        for i in range(8):
            d = 16 * s + 8 * h + i
            out[i] = x[row, d] if d < D else (1.0 if d == D else 0.0)
        return out

    a_units = {}                                       # (chunk, k-half, row) -> the 8 values the producing lane writes to the A stage
    for b in range(8):
        hacc = np.zeros((64, 16))
        for s in range(2):
            a = np.stack([w1_unit(b, s, l >> 5, l & 31) for l in range(64)])
            bb = np.stack([x_unit(l & 31, s, l >> 5) for l in range(64)])
            hacc = mfma(a, bb, hacc)
        for l in range(64):                            # lane = batch row, register r = feature 32 b + row_of_reg(r, h)
            for r in range(16):
                assert abs(hacc[l, r] - z1[l & 31, 32 * b + row_of_reg(r, l >> 5)]) < 1e-12
        e = elu(hacc)
        for l in range(64):
            for q in range(2):
                a_units[(2 * b + q, l >> 5, l & 31)] = e[l, 8 * q:8 * q + 8]
                for i in range(8):                     # ... which are exactly the features the launch's K order puts in that unit
                    assert abs(e[l, 8 * q + i] - h1[l & 31, feat(2 * b + q, l >> 5, i)]) < 1e-12
    # the h1 rows the kernel writes back: lane (row, h) stores registers 8 q .. + 3 at feature 32 b + 16 q + 4 h and 8 q + 4 .. + 7 at + 8
    for q in range(2):
        for h in range(2):
            assert [perm(h, i) for i in range(4)] == [4 * h + j for j in range(4)]
            assert [perm(h, i) for i in range(4, 8)] == [8 + 4 * h + j for j in range(4)]

    # main product, natural orientation: one 32 x 32 output tile per column block, K in 16 chunks of 16 slots
    for nt in range(2):                                # two of the eight column tiles are enough
        acc = np.zeros((64, 16))
        for c in range(16):
            a = np.stack([a_units[(c, l >> 5, l & 31)] for l in range(64)])                        # fragment read: (row, k-half)
            bfrag = np.stack([[W2[32 * nt + (l & 31), feat(c, l >> 5, i)] for i in range(8)] for l in range(64)])   # chain-ordered planes
            acc = mfma(a, bfrag, acc)
        for l in range(64):
            for r in range(16):
                assert abs(acc[l, r] - ref[row_of_reg(r, l >> 5), 32 * nt + (l & 31)]) < 1e-10
