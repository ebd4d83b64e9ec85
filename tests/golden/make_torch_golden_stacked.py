"""Golden vectors for recurrent chains with RnnBaseConfig::num_layers > 1 (src/torch/modules/seq/rnn/mod.rs:223-257),
computed with PyTorch on the CPU — the library the reference binds through tch.

    python tests/golden/make_torch_golden_stacked.py      -> tests/golden/torch_golden_stacked.json

Per case: Chain<Gru | Lstm (num_layers = L), Mlp> (modules/chain.rs:127-186) over lane trajectories with episode
boundaries — per-step outputs (torch.gru_cell / torch.lstm_cell layer by layer, relu, two Linear layers), successor
outputs at cut episodes, the gradient of sum(dout * out) through time by autograd, and the forward-mode derivative along
a random tangent (torch.autograd.functional.jvp).  The layer-by-layer stepping is asserted equal to torch.nn.GRU /
torch.nn.LSTM with num_layers = L over an unbroken sequence (what Tensor::gru / ::lstm compute, seq/rnn/gru.rs:41-66),
so the vectors carry libtorch's stacked-layer semantics.  Flat parameter order = trainable_variables(): per layer
[W_ih, W_hh, b_ih, b_hh], then W1, b1, W2, b2.
"""
import json
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def sizes_of(cell, D, H, L, H2, A):
    G = 4 if cell == "lstm" else 3
    s = []
    for l in range(L):
        K = D if l == 0 else H
        s += [G * H * K, G * H * H, G * H, G * H]
    return s + [H2 * H, H2, A * H2, A], G


def split(pv, cell, D, H, L, H2, A):
    sizes, G = sizes_of(cell, D, H, L, H2, A)
    parts, o = [], 0
    for sz in sizes:
        parts.append(pv[o:o + sz])
        o += sz
    layers = []
    for l in range(L):
        K = D if l == 0 else H
        wih, whh, bih, bhh = parts[4 * l:4 * l + 4]
        layers.append((wih.reshape(G * H, K), whh.reshape(G * H, H), bih, bhh))
    w1, b1, w2, b2 = parts[4 * L:]
    return layers, (w1.reshape(H2, H), b1, w2.reshape(A, H2), b2)


def step(cell, layers, x, state):
    """x [1][K]; state: list of h (GRU) or (h, c) (LSTM) per layer -> (top output, new state)"""
    new = []
    inp = x
    for l, (wih, whh, bih, bhh) in enumerate(layers):
        if cell == "lstm":
            h, c = torch.lstm_cell(inp, state[l], wih, whh, bih, bhh)
            new.append((h, c))
        else:
            h = torch.gru_cell(inp, state[l], wih, whh, bih, bhh)
            new.append(h)
        inp = h
    return inp, new


def zero_state(cell, L, H, dtype):
    z = lambda: torch.zeros(1, H, dtype=dtype)
    return [(z(), z()) if cell == "lstm" else z() for _ in range(L)]


def head_of(hd):
    w1, b1, w2, b2 = hd
    F = torch.nn.functional
    return lambda h: F.linear(torch.relu(F.linear(torch.relu(h), w1, b1)), w2, b2)


def check_against_nn_module(cell, layers, D, H, L, dtype, g):
    """the layer-by-layer stepping == torch.nn.GRU / LSTM(num_layers = L) on an unbroken sequence"""
    mod = (torch.nn.LSTM if cell == "lstm" else torch.nn.GRU)(D, H, num_layers=L, batch_first=True).to(dtype)
    with torch.no_grad():
        for l, (wih, whh, bih, bhh) in enumerate(layers):
            getattr(mod, f"weight_ih_l{l}").copy_(wih)
            getattr(mod, f"weight_hh_l{l}").copy_(whh)
            getattr(mod, f"bias_ih_l{l}").copy_(bih)
            getattr(mod, f"bias_hh_l{l}").copy_(bhh)
        xs = torch.randn(1, 6, D, generator=g, dtype=torch.float64).to(dtype)
        want, _ = mod(xs)
        st = zero_state(cell, L, H, dtype)
        for t in range(6):
            top, st = step(cell, layers, xs[:, t, :], st)
            assert torch.allclose(top, want[:, t, :], rtol=1e-5 if dtype == torch.float32 else 1e-12, atol=1e-6 if dtype == torch.float32 else 1e-13)


def stacked_case(cell, dtype, D=3, H=4, L=2, H2=5, A=2, n=3, T=7, seed=51):
    g = torch.Generator().manual_seed(seed)
    sizes, G = sizes_of(cell, D, H, L, H2, A)
    P = sum(sizes)
    params = ((torch.rand(P, generator=g, dtype=torch.float64) * 2 - 1) * 0.7).to(dtype).requires_grad_(True)
    obs = torch.randn(D, T + 1, n, generator=g, dtype=torch.float64).to(dtype)
    term_obs = torch.randn(D, T, n, generator=g, dtype=torch.float64).to(dtype)
    flag = torch.zeros(T, n, dtype=torch.int64)
    flag[2, 0] = 1   # Terminate
    flag[4, 0] = 2   # Interrupt
    flag[3, 1] = 2
    flag[T - 1, 2] = 1
    dout = torch.randn(A, T, n, generator=g, dtype=torch.float64).to(dtype)
    layers, hd = split(params, cell, D, H, L, H2, A)
    check_against_nn_module(cell, [tuple(a.detach() for a in w) for w in layers], D, H, L, dtype, g)
    head = head_of(hd)
    out = torch.zeros(A, T, n, dtype=dtype)
    succ = torch.zeros(A, T, n, dtype=dtype)
    outs = []
    for i in range(n):
        st = zero_state(cell, L, H, dtype)
        for t in range(T):
            top, st = step(cell, layers, obs[:, t, i].unsqueeze(0), st)
            outs.append((i, t, head(top).squeeze(0)))
            f = int(flag[t, i])
            if f == 2 or (f == 0 and t == T - 1):
                xs = term_obs[:, t, i] if f == 2 else obs[:, T, i]
                tops, _ = step(cell, layers, xs.unsqueeze(0), st)
                succ[:, t, i] = head(tops).squeeze(0).detach()
            if f != 0:
                st = zero_state(cell, L, H, dtype)
    loss = 0
    for i, t, y in outs:
        out[:, t, i] = y.detach()
        loss = loss + (dout[:, t, i] * y).sum()
    loss.backward()
    tangent = torch.randn(P, generator=g, dtype=torch.float64).to(dtype)

    def all_outputs(pv):
        ly, h2 = split(pv, cell, D, H, L, H2, A)
        hf = head_of(h2)
        ys = []
        for i in range(n):
            st = zero_state(cell, L, H, dtype)
            for t in range(T):
                top, st = step(cell, ly, obs[:, t, i].unsqueeze(0), st)
                ys.append(hf(top).squeeze(0))
                if int(flag[t, i]) != 0:
                    st = zero_state(cell, L, H, dtype)
        return torch.stack(ys)  # [n * T, A]

    _, jv = torch.autograd.functional.jvp(all_outputs, params.detach(), tangent)
    out_dot = jv.reshape(n, T, A).permute(2, 1, 0).contiguous()
    return {
        "cell": cell, "dims": [D, H, L, H2, A], "n": n, "T": T, "dtype": str(dtype).replace("torch.", ""),
        "params": params.detach().double().tolist(), "obs": obs.double().flatten().tolist(),
        "term_obs": term_obs.double().flatten().tolist(), "flag": flag.flatten().tolist(),
        "dout": dout.double().flatten().tolist(), "out": out.double().flatten().tolist(),
        "succ_out": succ.double().flatten().tolist(), "grad": params.grad.double().tolist(),
        "tangent": tangent.double().tolist(), "out_dot": out_dot.double().flatten().tolist(),
    }


def main():
    torch.set_num_threads(1)
    cases = {"generator": f"torch {torch.__version__} (CPU), make_torch_golden_stacked.py"}
    for cell in ("gru", "lstm"):
        cases[f"{cell}_l2_f64"] = stacked_case(cell, torch.float64, L=2, seed=51)
        cases[f"{cell}_l2_f32"] = stacked_case(cell, torch.float32, L=2, seed=51)
        cases[f"{cell}_l3_f64"] = stacked_case(cell, torch.float64, D=2, H=5, L=3, H2=3, A=1, n=4, T=6, seed=53)
    with open(os.path.join(HERE, "torch_golden_stacked.json"), "w") as f:
        json.dump(cases, f)
    print("wrote torch_golden_stacked.json")


if __name__ == "__main__":
    main()
