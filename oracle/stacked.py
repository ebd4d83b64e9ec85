"""Stacked recurrent layers -> ReLU -> MLP in NumPy f64: forward with a record, backward through time, forward-mode
derivative.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the checker for the device's stacked-layer path
(relearn_amd/csrc/kernels_seq_stack.hip).  Restates what the reference gets from libtorch for
``RnnBaseConfig { num_layers > 1 }`` (src/torch/modules/seq/rnn/mod.rs:223-257 for the weight order
[W_ih, W_hh, b_ih, b_hh] per layer; seq/rnn/gru.rs:41-66 / lstm.rs for ``Tensor::gru`` / ``::lstm`` with
``num_layers``; modules/chain.rs:127-186 for the ReLU + Mlp head).  libtorch's cells (third party, public semantics):

    GRU   r = s(W_hr h + b_hr + W_ir x + b_ir), z likewise, n = tanh(W_in x + b_in + r (W_hn h + b_hn)),
          h' = (h - n) z + n                                         gate rows [r; z; n]
    LSTM  i, f, o = s(.), g = tanh(.) of W_h. h + b_h. + W_i. x + b_i., c' = f c + i g, h' = o tanh(c')
                                                                     gate rows [i; f; g; o]

Layer l > 0 reads the hidden output of layer l - 1 at the same step; every layer's state is zero at t = 0 and after a
step whose flag ends the episode.  Pinned by tests/test_stacked_oracle.py against torch.nn.GRU / torch.nn.LSTM
(num_layers, f64, autograd) on the CPU — the library the reference binds through tch — and against the C restatement of
the forward (oracle/stack_impl.inc).

Layouts are the lane trajectory's: obs [D][T+1][n], flag [T][n], term_obs [D][T][n]; outputs [A][T][n].
"""
import numpy as np

CONTINUE, TERMINATE, INTERRUPT = 0, 1, 2
GRU, LSTM = 0, 1


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


class Spec:
    def __init__(self, cell, in_dim, hidden, num_layers, mlp_hidden, out_dim):
        self.cell, self.D, self.H, self.L, self.H2, self.A = cell, in_dim, hidden, num_layers, mlp_hidden, out_dim
        self.G = 4 if cell == LSTM else 3

    def slices(self):
        """[(name, layer, shape, offset)] in flat order"""
        out, o = [], 0
        GH = self.G * self.H
        for l in range(self.L):
            K = self.D if l == 0 else self.H
            for name, shape in (("Wih", (GH, K)), ("Whh", (GH, self.H)), ("bih", (GH,)), ("bhh", (GH,))):
                out.append((name, l, shape, o))
                o += int(np.prod(shape))
        for name, shape in (("W1", (self.H2, self.H)), ("b1", (self.H2,)), ("W2", (self.A, self.H2)), ("b2", (self.A,))):
            out.append((name, -1, shape, o))
            o += int(np.prod(shape))
        return out, o

    def num_params(self):
        return self.slices()[1]

    def unpack(self, flat):
        flat = np.asarray(flat, dtype=np.float64)
        sl, P = self.slices()
        assert flat.shape == (P,)
        layers = [dict() for _ in range(self.L)]
        head = {}
        for name, l, shape, o in sl:
            (head if l < 0 else layers[l])[name] = flat[o:o + int(np.prod(shape))].reshape(shape)
        return layers, head

    def pack(self, layers, head):
        sl, P = self.slices()
        flat = np.zeros(P)
        for name, l, shape, o in sl:
            flat[o:o + int(np.prod(shape))] = (head if l < 0 else layers[l])[name].reshape(-1)
        return flat


def _cell(spec, w, x, h, c):
    """one layer, all lanes: x [n][K], h / c [n][H] -> (h', c', record)"""
    H = spec.H
    gi = x @ w["Wih"].T + w["bih"]
    gh = h @ w["Whh"].T + w["bhh"]
    if spec.cell == LSTM:
        pre = gh + gi
        i, f, g, o = _sig(pre[:, :H]), _sig(pre[:, H:2 * H]), np.tanh(pre[:, 2 * H:3 * H]), _sig(pre[:, 3 * H:])
        cn = f * c + i * g
        tc = np.tanh(cn)
        return o * tc, cn, dict(x=x, h=h, c=c, i=i, f=f, g=g, o=o, tc=tc)
    r = _sig(gh[:, :H] + gi[:, :H])
    z = _sig(gh[:, H:2 * H] + gi[:, H:2 * H])
    ghn = gh[:, 2 * H:]
    nn = np.tanh(gi[:, 2 * H:] + r * ghn)
    return (h - nn) * z + nn, c, dict(x=x, h=h, r=r, z=z, n=nn, ghn=ghn)


def _step(spec, layers, head, x, hs, cs):
    """all layers + head from the states hs / cs (lists of [n][H]); -> (out [n][A], new hs, new cs, record)"""
    rec = {"cells": []}
    inp = x
    nh, nc = [], []
    for l in range(spec.L):
        hn, cn, r = _cell(spec, layers[l], inp, hs[l], cs[l])
        rec["cells"].append(r)
        nh.append(hn)
        nc.append(cn)
        inp = hn
    a1 = np.maximum(inp, 0.0)
    u = np.maximum(a1 @ head["W1"].T + head["b1"], 0.0)
    rec.update(top=inp, a1=a1, u=u)
    return u @ head["W2"].T + head["b2"], nh, nc, rec


def forward(spec, params, traj, want_succ=True, keep_record=False):
    """-> (out [A][T][n], succ [A][T][n] or None, record or None)"""
    layers, head = spec.unpack(params)
    obs = np.asarray(traj["obs"], dtype=np.float64)
    flag = np.asarray(traj["flag"])
    D, T1, n = obs.shape
    T = T1 - 1
    out = np.zeros((spec.A, T, n))
    succ = np.zeros((spec.A, T, n)) if want_succ else None
    hs = [np.zeros((n, spec.H)) for _ in range(spec.L)]
    cs = [np.zeros((n, spec.H)) for _ in range(spec.L)]
    record = []
    for t in range(T):
        o, nh, nc, rec = _step(spec, layers, head, obs[:, t, :].T, hs, cs)
        out[:, t, :] = o.T
        if keep_record:
            record.append(rec)
        f = flag[t]
        if want_succ:
            need = (f == INTERRUPT) | ((f == CONTINUE) & (t == T - 1))
            if need.any():
                term = np.asarray(traj["term_obs"], dtype=np.float64)[:, t, :].T
                x2 = np.where((f == INTERRUPT)[:, None], term, obs[:, T, :].T)
                o2, _, _, _ = _step(spec, layers, head, x2, nh, nc)  # the states are not advanced
                succ[:, t, :] = np.where(need[None, :], o2.T, 0.0)
        live = (f == CONTINUE)[:, None]
        hs = [np.where(live, a, 0.0) for a in nh]
        cs = [np.where(live, a, 0.0) for a in nc]
    return out, succ, (record if keep_record else None)


def backward(spec, params, traj, dout):
    """sum over (a, t, lane) of dout[a][t][lane] * d out[a][t][lane] / d params  (flat, f64)"""
    layers, head = spec.unpack(params)
    flag = np.asarray(traj["flag"])
    _, _, record = forward(spec, params, traj, want_succ=False, keep_record=True)
    dout = np.asarray(dout, dtype=np.float64)
    A, T, n = dout.shape
    H = spec.H
    gl = [{k: np.zeros_like(v) for k, v in w.items()} for w in layers]
    gh_ = {k: np.zeros_like(v) for k, v in head.items()}
    dh = [np.zeros((n, H)) for _ in range(spec.L)]  # gradients flowing into (h', c') of step t from step t + 1
    dc = [np.zeros((n, H)) for _ in range(spec.L)]
    for t in range(T - 1, -1, -1):
        rec = record[t]
        ended = (flag[t] != CONTINUE)[:, None]  # the next step starts a new episode
        dh = [np.where(ended, 0.0, a) for a in dh]
        dc = [np.where(ended, 0.0, a) for a in dc]
        d = dout[:, t, :].T  # [n][A]
        gh_["b2"] += d.sum(0)
        gh_["W2"] += d.T @ rec["u"]
        du = (d @ head["W2"]) * (rec["u"] > 0)
        gh_["b1"] += du.sum(0)
        gh_["W1"] += du.T @ rec["a1"]
        dtop = (du @ head["W1"]) * (rec["top"] > 0)
        din = dtop  # gradient into the output of the layer below the one being processed (same step)
        for l in range(spec.L - 1, -1, -1):
            c = rec["cells"][l]
            w = layers[l]
            dhl = dh[l] + din
            if spec.cell == LSTM:
                dO = dhl * c["tc"]
                dcn = dc[l] + dhl * c["o"] * (1.0 - c["tc"] ** 2)
                dI, dF, dG = dcn * c["g"], dcn * c["c"], dcn * c["i"]
                dc[l] = dcn * c["f"]
                dgi = np.concatenate([dI * c["i"] * (1 - c["i"]), dF * c["f"] * (1 - c["f"]), dG * (1 - c["g"] ** 2),
                                      dO * c["o"] * (1 - c["o"])], axis=1)
                dgh = dgi
                dh[l] = dgh @ w["Whh"]
            else:
                dz = dhl * (c["h"] - c["n"])
                dn = dhl * (1.0 - c["z"])
                dpn = dn * (1.0 - c["n"] ** 2)
                dr = dpn * c["ghn"]
                dgr = dr * c["r"] * (1 - c["r"])
                dgz = dz * c["z"] * (1 - c["z"])
                dgi = np.concatenate([dgr, dgz, dpn], axis=1)
                dgh = np.concatenate([dgr, dgz, dpn * c["r"]], axis=1)
                dh[l] = dhl * c["z"] + dgh @ w["Whh"]
            gl[l]["bih"] += dgi.sum(0)
            gl[l]["bhh"] += dgh.sum(0)
            gl[l]["Wih"] += dgi.T @ c["x"]
            gl[l]["Whh"] += dgh.T @ c["h"]
            din = dgi @ w["Wih"]
    return spec.pack(gl, gh_)


def jvp(spec, params, tangent, traj):
    """forward-mode derivative: d out[a][t][lane] / d params . tangent  -> [A][T][n]"""
    layers, head = spec.unpack(params)
    vl, vh = spec.unpack(tangent)
    flag = np.asarray(traj["flag"])
    _, _, record = forward(spec, params, traj, want_succ=False, keep_record=True)
    T, n, H = len(record), flag.shape[1], spec.H
    out = np.zeros((spec.A, T, n))
    hd = [np.zeros((n, H)) for _ in range(spec.L)]
    cd = [np.zeros((n, H)) for _ in range(spec.L)]
    for t in range(T):
        rec = record[t]
        ind = None  # tangent of the layer input (the observations carry none)
        for l in range(spec.L):
            c, w, v = rec["cells"][l], layers[l], vl[l]
            gid = c["x"] @ v["Wih"].T + v["bih"]
            if ind is not None:
                gid = gid + ind @ w["Wih"].T
            ghd = c["h"] @ v["Whh"].T + v["bhh"] + hd[l] @ w["Whh"].T
            if spec.cell == LSTM:
                pd = gid + ghd
                i_d = c["i"] * (1 - c["i"]) * pd[:, :H]
                f_d = c["f"] * (1 - c["f"]) * pd[:, H:2 * H]
                g_d = (1 - c["g"] ** 2) * pd[:, 2 * H:3 * H]
                o_d = c["o"] * (1 - c["o"]) * pd[:, 3 * H:]
                cn_d = f_d * c["c"] + c["f"] * cd[l] + i_d * c["g"] + c["i"] * g_d
                tc_d = (1 - c["tc"] ** 2) * cn_d
                cd[l] = cn_d
                hd[l] = o_d * c["tc"] + c["o"] * tc_d
            else:
                r_d = c["r"] * (1 - c["r"]) * (gid[:, :H] + ghd[:, :H])
                z_d = c["z"] * (1 - c["z"]) * (gid[:, H:2 * H] + ghd[:, H:2 * H])
                n_d = (1 - c["n"] ** 2) * (gid[:, 2 * H:] + r_d * c["ghn"] + c["r"] * ghd[:, 2 * H:])
                hd[l] = (hd[l] - n_d) * c["z"] + (c["h"] - c["n"]) * z_d + n_d
            ind = hd[l]
        a1d = ind * (rec["top"] > 0)
        ud = (rec["a1"] @ vh["W1"].T + vh["b1"] + a1d @ head["W1"].T) * (rec["u"] > 0)
        out[:, t, :] = (rec["u"] @ vh["W2"].T + vh["b2"] + ud @ head["W2"].T).T
        live = (flag[t] == CONTINUE)[:, None]
        hd = [np.where(live, a, 0.0) for a in hd]
        cd = [np.where(live, a, 0.0) for a in cd]
    return out


def policy_fvp(spec, params, v, traj, reg):
    """Fisher-vector product of the mean KL at params: J^T (diag(p) - p p^T) J v / B + reg v (the composition of
    oracle.gru_policy_fvp)"""
    logits, _, _ = forward(spec, params, traj, want_succ=False)
    od = jvp(spec, params, v, traj)
    z = logits - logits.max(0)
    pr = np.exp(z - np.log(np.exp(z).sum(0)))
    B = logits[0].size
    pdz = (pr * od).sum(0)
    dz = pr * (od - pdz) / B
    return backward(spec, params, traj, dz) + reg * np.asarray(v, dtype=np.float64)
