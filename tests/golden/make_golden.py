#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- golden input/output vectors for the LRCN hot path.

The reference is Julia/Knet and cannot run in this container (no julia), so these vectors
are NOT outputs of the reference: they come from an independent torch-CPU float64 *autograd* transcription of
lrcn.jl written below (forward only is transcribed; every gradient is torch autograd's, not hand-derived), and
serve to pin the C oracle (oracle/lrcn_oracle.c, hand-derived backward) and the HIP path against a second
statement of the same algorithm.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
EOS, BOS, UNK = 0, 1, 2
torch.set_default_dtype(torch.float64)


def xavier(rng, rows, cols):
    s = np.sqrt(2.0 / (rows + cols))
    return (2 * s * rng.random((rows, cols)) - s).astype(np.float32)


def init_model(rng, E, H1, H2, V, F=4096):
    h = (H2 + 1) // 2
    b1 = np.zeros((1, 4 * H1), np.float32)
    b1[0, :H1] = 1
    b2 = np.zeros((1, 4 * H2), np.float32)
    b2[0, :H2] = 1
    return {"W1": xavier(rng, E + H1, 4 * H1), "b1": b1, "W2": xavier(rng, 2 * H2, 4 * H2), "b2": b2,
            "Wproj": xavier(rng, H1, h), "Wcnn": xavier(rng, F, h), "Wembed": xavier(rng, V, E),
            "Wout": xavier(rng, H2, V), "bout": (0.1 * rng.standard_normal((1, V))).astype(np.float32)}


def lstm(W, b, h, c, x):  # lrcn.jl:528-538
    gates = torch.cat([x, h], 1) @ W + b
    H = h.shape[1]
    f = torch.sigmoid(gates[:, :H])
    i = torch.sigmoid(gates[:, H:2 * H])
    o = torch.sigmoid(gates[:, 2 * H:3 * H])
    g = torch.tanh(gates[:, 3 * H:])
    c = c * f + i * g
    h = o * torch.tanh(c)
    return h, c


def lrcn(p, s, x_cnn, x_lstm, m1=None, m2=None):  # lrcn.jl:540-551
    x = x_lstm if m1 is None else x_lstm * m1
    s[0], s[1] = lstm(p["W1"], p["b1"], s[0], s[1], x)
    x = s[0] @ p["Wproj"]
    x = torch.cat([x, x_cnn], 1)
    if m2 is not None:
        x = x * m2
    s[2], s[3] = lstm(p["W2"], p["b2"], s[2], s[3], x)
    return s[2] @ p["Wout"] + p["bout"]


def loss(p, feats, tokens, norm_B, mask1=None, mask2=None, collect=None):  # lrcn.jl:553-581
    T, B = tokens.shape
    H1 = p["Wproj"].shape[0]
    H2 = p["Wout"].shape[0]
    s = [torch.zeros(B, H1), torch.zeros(B, H1), torch.zeros(B, H2), torch.zeros(B, H2)]
    total = 0.0
    count = 0
    x_lstm = p["Wembed"][torch.full((B,), BOS, dtype=torch.long)]
    x_cnn = feats @ p["Wcnn"]
    for t in range(T + 1):
        ypred = lrcn(p, s, x_cnn, x_lstm, None if mask1 is None else mask1[t], None if mask2 is None else mask2[t])
        if collect is not None:
            collect.append(ypred.detach().numpy().astype(np.float32))
        ynorm = torch.log_softmax(ypred, 1)
        tgt = torch.as_tensor(tokens[t], dtype=torch.long) if t < T else torch.full((B,), EOS, dtype=torch.long)
        total = total + ynorm[torch.arange(B), tgt].sum()
        count += norm_B
        if t < T:
            x_lstm = p["Wembed"][tgt]
    return -total / count


def adam_ref(w, g, m, v, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):  # Knet Adam defaults
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    w = w - lr * (m / (1 - b1 ** t)) / (np.sqrt(v / (1 - b2 ** t)) + eps)
    return w, m, v


def beam_search_ref(p, feat, K, nword):  # lrcn.jl:585-678 / SURVEY A.3, float32 probabilities like the reference
    with torch.no_grad():
        H1 = p["Wproj"].shape[0]
        H2 = p["Wout"].shape[0]
        x_cnn = feat @ p["Wcnn"]
        x = [([BOS], np.float32(1.0)) for _ in range(K)]
        states = [[torch.zeros(1, H1), torch.zeros(1, H1), torch.zeros(1, H2), torch.zeros(1, H2)] for _ in range(K)]
        current = 1
        while True:
            new_x = []
            for i in range(K):
                last = x[i][0][-1]
                yp = lrcn(p, states[i], x_cnn, p["Wembed"][last:last + 1])
                prob = torch.softmax(yp, 1).numpy().astype(np.float32).reshape(-1)
                top = np.argsort(-prob, kind="stable")[:K]
                for j in range(K):
                    new_x.append((x[i][0] + [int(top[j])], np.float32(prob[top[j]] * x[i][1])))
                if current == 1:
                    break
            order = np.argsort(-np.array([c[1] for c in new_x], np.float32), kind="stable")
            xs = [new_x[o] for o in order[:K]]
            if xs[0][0][-1] == EOS or current > nword:
                return xs
            states = [[t.clone() for t in states[order[i] // K]] for i in range(K)]
            x = xs
            current += 1


def make_lstm_case(name, seed, B, E, H1, H2, V, T, pdrop, norm_B=None, nadam=2, beam=None):
    rng = np.random.default_rng(seed)
    norm_B = norm_B or B
    P = init_model(rng, E, H1, H2, V)
    feats = (rng.standard_normal((B, 4096)) * 0.05).astype(np.float32)
    tokens = rng.integers(0, V, size=(T, B)).astype(np.int32)
    mask1 = mask2 = None
    if pdrop > 0:
        mask1 = ((rng.random((T + 1, B, E)) > pdrop) / (1 - pdrop)).astype(np.float32)
        mask2 = ((rng.random((T + 1, B, H2)) > pdrop) / (1 - pdrop)).astype(np.float32)
    p = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in P.items()}
    tm1 = None if mask1 is None else torch.tensor(mask1, dtype=torch.float64)
    tm2 = None if mask2 is None else torch.tensor(mask2, dtype=torch.float64)
    logits = []
    L = loss(p, torch.tensor(feats, dtype=torch.float64), tokens, norm_B, tm1, tm2, collect=logits)
    L.backward()
    out = {"E": E, "H1": H1, "H2": H2, "V": V, "T": T, "B": B, "norm_B": norm_B, "pdrop": pdrop,
           "feats": feats, "tokens": tokens, "loss": np.float64(L.item()), "logits": np.stack(logits)}
    if mask1 is not None:
        out["mask1"], out["mask2"] = mask1, mask2
    for k in P:
        out["p_" + k] = P[k]
        out["g_" + k] = p[k].grad.numpy().astype(np.float32)
    # a few Adam steps on the same batch (update!, lrcn.jl:394): params after `nadam` steps
    W = {k: v.astype(np.float64) for k, v in P.items()}
    M = {k: np.zeros_like(v) for k, v in W.items()}
    Vv = {k: np.zeros_like(v) for k, v in W.items()}
    losses = []
    for t in range(1, nadam + 1):
        q = {k: torch.tensor(v.astype(np.float32), dtype=torch.float64, requires_grad=True) for k, v in W.items()}
        Lt = loss(q, torch.tensor(feats, dtype=torch.float64), tokens, norm_B, tm1, tm2)
        Lt.backward()
        losses.append(Lt.item())
        for k in W:
            W[k], M[k], Vv[k] = adam_ref(W[k].astype(np.float32).astype(np.float64), q[k].grad.numpy(), M[k], Vv[k], t)
    out["adam_losses"] = np.array(losses)
    for k in W:
        out["a_" + k] = W[k].astype(np.float32)
    if beam:
        K, nword = beam
        bs = []
        for i in range(min(B, 4)):
            xs = beam_search_ref({k: v.detach() for k, v in p.items()},
                                 torch.tensor(feats[i:i + 1], dtype=torch.float64), K, nword)
            seq = np.full(nword + 3, -1, np.int32)
            seq[:len(xs[0][0])] = xs[0][0]
            bs.append(seq)
            out.setdefault("beam_prob", []).append(xs[0][1])
        out["beam_tokens"] = np.stack(bs)
        out["beam_prob"] = np.array(out["beam_prob"], np.float32)
        out["beam_K"], out["beam_nword"] = K, nword
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", L.item())


# ---- LRCN-1f (BASELINE configs[1] "1-layer LSTM"; this repo's definition, see oracle/lrcn_oracle.h): one lstm() over
# dropout(hcat(x_lstm, x_cnn)), then the reference's output layer; loss / Adam / beam search are the reference's code around it.
def init_model1(rng, E, H, V, F=4096):
    h = (H + 1) // 2
    b1 = np.zeros((1, 4 * H), np.float32)
    b1[0, :H] = 1
    z = np.zeros((0, 0), np.float32)
    return {"W1": xavier(rng, E + h + H, 4 * H), "b1": b1, "W2": z, "b2": z, "Wproj": z, "Wcnn": xavier(rng, F, h),
            "Wembed": xavier(rng, V, E), "Wout": xavier(rng, H, V), "bout": (0.1 * rng.standard_normal((1, V))).astype(np.float32)}


def lrcn1(p, s, x_cnn, x_lstm, m=None):
    x = torch.cat([x_lstm, x_cnn], 1)
    if m is not None:
        x = x * m
    s[0], s[1] = lstm(p["W1"], p["b1"], s[0], s[1], x)
    return s[0] @ p["Wout"] + p["bout"]


def loss1(p, feats, tokens, norm_B, mask=None, collect=None):  # lrcn.jl:553-581 around lrcn1
    T, B = tokens.shape
    H = p["Wout"].shape[0]
    s = [torch.zeros(B, H), torch.zeros(B, H)]
    total = 0.0
    count = 0
    x_lstm = p["Wembed"][torch.full((B,), BOS, dtype=torch.long)]
    x_cnn = feats @ p["Wcnn"]
    for t in range(T + 1):
        ypred = lrcn1(p, s, x_cnn, x_lstm, None if mask is None else mask[t])
        if collect is not None:
            collect.append(ypred.detach().numpy().astype(np.float32))
        ynorm = torch.log_softmax(ypred, 1)
        tgt = torch.as_tensor(tokens[t], dtype=torch.long) if t < T else torch.full((B,), EOS, dtype=torch.long)
        total = total + ynorm[torch.arange(B), tgt].sum()
        count += norm_B
        if t < T:
            x_lstm = p["Wembed"][tgt]
    return -total / count


def beam_search_ref1(p, feat, K, nword):
    with torch.no_grad():
        H = p["Wout"].shape[0]
        x_cnn = feat @ p["Wcnn"]
        x = [([BOS], np.float32(1.0)) for _ in range(K)]
        states = [[torch.zeros(1, H), torch.zeros(1, H)] for _ in range(K)]
        current = 1
        while True:
            new_x = []
            for i in range(K):
                last = x[i][0][-1]
                yp = lrcn1(p, states[i], x_cnn, p["Wembed"][last:last + 1])
                prob = torch.softmax(yp, 1).numpy().astype(np.float32).reshape(-1)
                top = np.argsort(-prob, kind="stable")[:K]
                for j in range(K):
                    new_x.append((x[i][0] + [int(top[j])], np.float32(prob[top[j]] * x[i][1])))
                if current == 1:
                    break
            order = np.argsort(-np.array([c[1] for c in new_x], np.float32), kind="stable")
            xs = [new_x[o] for o in order[:K]]
            if xs[0][0][-1] == EOS or current > nword:
                return xs
            states = [[t.clone() for t in states[order[i] // K]] for i in range(K)]
            x = xs
            current += 1


def make_lstm1_case(name, seed, B, E, H, V, T, pdrop, norm_B=None, nadam=2, beam=None):
    rng = np.random.default_rng(seed)
    norm_B = norm_B or B
    h = (H + 1) // 2
    P = init_model1(rng, E, H, V)
    live = [k for k in P if P[k].size]
    feats = (rng.standard_normal((B, 4096)) * 0.05).astype(np.float32)
    tokens = rng.integers(0, V, size=(T, B)).astype(np.int32)
    mask = ((rng.random((T + 1, B, E + h)) > pdrop) / (1 - pdrop)).astype(np.float32) if pdrop > 0 else None
    tm = None if mask is None else torch.tensor(mask, dtype=torch.float64)
    p = {k: torch.tensor(P[k], dtype=torch.float64, requires_grad=True) for k in live}
    logits = []
    L = loss1(p, torch.tensor(feats, dtype=torch.float64), tokens, norm_B, tm, collect=logits)
    L.backward()
    out = {"E": E, "H1": H, "H2": H, "V": V, "T": T, "B": B, "norm_B": norm_B, "pdrop": pdrop, "n_layers": 1,
           "feats": feats, "tokens": tokens, "loss": np.float64(L.item()), "logits": np.stack(logits)}
    if mask is not None:
        out["mask1"] = mask
    for k in P:
        out["p_" + k] = P[k]
        out["g_" + k] = p[k].grad.numpy().astype(np.float32) if k in live else P[k]
    W = {k: P[k].astype(np.float64) for k in live}
    M = {k: np.zeros_like(v) for k, v in W.items()}
    Vv = {k: np.zeros_like(v) for k, v in W.items()}
    losses = []
    for t in range(1, nadam + 1):
        q = {k: torch.tensor(v.astype(np.float32), dtype=torch.float64, requires_grad=True) for k, v in W.items()}
        Lt = loss1(q, torch.tensor(feats, dtype=torch.float64), tokens, norm_B, tm)
        Lt.backward()
        losses.append(Lt.item())
        for k in W:
            W[k], M[k], Vv[k] = adam_ref(W[k].astype(np.float32).astype(np.float64), q[k].grad.numpy(), M[k], Vv[k], t)
    out["adam_losses"] = np.array(losses)
    for k in P:
        out["a_" + k] = W[k].astype(np.float32) if k in live else P[k]
    if beam:
        K, nword = beam
        bs = []
        for i in range(min(B, 4)):
            xs = beam_search_ref1({k: v.detach() for k, v in p.items()}, torch.tensor(feats[i:i + 1], dtype=torch.float64), K, nword)
            seq = np.full(nword + 3, -1, np.int32)
            seq[:len(xs[0][0])] = xs[0][0]
            bs.append(seq)
            out.setdefault("beam_prob", []).append(xs[0][1])
        out["beam_tokens"] = np.stack(bs)
        out["beam_prob"] = np.array(out["beam_prob"], np.float32)
        out["beam_K"], out["beam_nword"] = K, nword
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", L.item())


def make_cnn_case(name, seed):
    """conv3x3(pad1, cross-correlation)+bias+relu, 2x2 max-pool, fc -- torch.nn.functional as the second statement.
    Julia (W,H,C,N) column-major == torch [N][C][H][W] with Julia dim 1 <-> torch W."""
    import torch.nn.functional as F
    rng = np.random.default_rng(seed)
    N, Cin, Cout, S = 2, 5, 7, 12
    x = rng.standard_normal((N, Cin, S, S)).astype(np.float32)          # torch NCHW
    w = (rng.standard_normal((Cout, Cin, 3, 3)) * 0.3).astype(np.float32)  # torch OIHW
    b = rng.standard_normal(Cout).astype(np.float32)
    y = F.relu(F.conv2d(torch.tensor(x, dtype=torch.float64), torch.tensor(w, dtype=torch.float64),
                        torch.tensor(b, dtype=torch.float64), padding=1))
    yp = F.max_pool2d(y, 2)
    K = Cout * (S // 2) ** 2
    O = 11
    w6 = (rng.standard_normal((O, K)) * 0.1).astype(np.float32)
    b6 = rng.standard_normal(O).astype(np.float32)
    f = torch.tensor(w6, dtype=torch.float64) @ yp.reshape(N, K).T + torch.tensor(b6, dtype=torch.float64)[:, None]
    np.savez_compressed(os.path.join(HERE, name + ".npz"),
                        # stored in the reference's axis order (W,H,C,N) / (3,3,Cin,Cout): reverse torch's axes
                        x=np.transpose(x, (3, 2, 1, 0)), w=np.transpose(w, (3, 2, 1, 0)), b=b,
                        y=np.transpose(y.numpy().astype(np.float32), (3, 2, 1, 0)),
                        yp=np.transpose(yp.numpy().astype(np.float32), (3, 2, 1, 0)),
                        w6=w6, b6=b6, f6=f.numpy().astype(np.float32))
    print(name, "ok")


if __name__ == "__main__":
    import sys
    only = set(sys.argv[1:])  # e.g. `make_golden.py lstm1_tiny lstm1_drop lstm1_mid` regenerates just those files

    def want(n):
        return not only or n in only

    if want("lstm_tiny"):
        make_lstm_case("lstm_tiny", 1, B=4, E=8, H1=8, H2=8, V=17, T=5, pdrop=0.0, beam=(3, 6))
    if want("lstm_tiny_drop"):
        make_lstm_case("lstm_tiny_drop", 2, B=4, E=8, H1=8, H2=8, V=17, T=5, pdrop=0.4)
    if want("lstm_ragged"):
        make_lstm_case("lstm_ragged", 3, B=3, E=12, H1=20, H2=10, V=23, T=1, pdrop=0.0, norm_B=6, beam=(4, 5))
    if want("lstm_mid"):
        make_lstm_case("lstm_mid", 4, B=16, E=40, H1=48, H2=56, V=203, T=9, pdrop=0.0, nadam=1, beam=(5, 12))
    if want("cnn_small"):
        make_cnn_case("cnn_small", 5)
    if want("lstm1_tiny"):
        make_lstm1_case("lstm1_tiny", 11, B=4, E=8, H=8, V=17, T=5, pdrop=0.0, beam=(3, 6))
    if want("lstm1_drop"):
        make_lstm1_case("lstm1_drop", 12, B=3, E=12, H=10, V=23, T=3, pdrop=0.4, norm_B=6)
    if want("lstm1_mid"):
        make_lstm1_case("lstm1_mid", 13, B=16, E=40, H=56, V=203, T=9, pdrop=0.0, nadam=1, beam=(5, 12))
