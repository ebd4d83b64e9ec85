#!/usr/bin/env python3
"""Generate tests/golden/torch_golden.json with PyTorch CPU autograd (run in the build container only).

The reference delegates its tensor math to libtorch 1.12 through `tch`; that source is not under
/root/reference.  PyTorch 2.10 (CPU) is importable in the build container and implements the same public
op semantics, so it is used here as an INDEPENDENT cross-check of the oracle's hand-derived math: the
quantities below are produced by real reverse-mode autograd, the Hessian-vector product by the same
double-backward construction the reference uses (HessianVectorProduct, conjugate_gradient.rs:262-339),
the TRPO step by a transcription of trust_region_backward_step / backtracking_line_search
(conjugate_gradient.rs:115-255) onto torch tensors, and Adam by torch.optim.Adam.

Only numbers are committed (inputs + expected outputs); this script is the generator.
"""
import json
import math
import os

import torch

torch.manual_seed(1234)
HERE = os.path.dirname(os.path.abspath(__file__))


def mlp_forward(params, x, D, H, A):
    W1 = params[:H * D].reshape(H, D)
    b1 = params[H * D:H * D + H]
    W2 = params[H * D + H:H * D + H + A * H].reshape(A, H)
    b2 = params[H * D + H + A * H:]
    h = torch.relu(torch.nn.functional.linear(x, W1, b1))
    return torch.nn.functional.linear(h, W2, b2)


def clamp_min_float(x):
    return x.clamp_min(torch.finfo(x.dtype).min)


def categorical_kl(lp_self, lp_other):
    return (clamp_min_float(lp_self - lp_other) * lp_self.exp()).sum(-1)


def categorical_entropy(lp):
    return -(clamp_min_float(lp) * lp.exp()).sum(-1)


def loss_distance(params, x, actions, adv, lp0, lp0_a, dims):
    lp = torch.log_softmax(mlp_forward(params, x, *dims), -1)
    lpa = lp.gather(-1, actions.unsqueeze(-1)).squeeze(-1)
    loss = -((lpa - lp0_a).exp() * adv).mean()
    dist = categorical_kl(lp0, lp).mean()
    return loss, dist


def solve_cg(f_Ax, b, iterations, tol):
    x = torch.zeros_like(b)
    r = b.clone()
    p = b.clone()
    rr = r.dot(r)
    iters = 0
    for _ in range(iterations):
        z = f_Ax(p)
        iters += 1
        alpha = rr / p.dot(z)
        x = x + alpha * p
        r = r - alpha * z
        new_rr = r.dot(r)
        if float(new_rr) < tol:
            break
        mu = new_rr / rr
        p = p * mu + r
        rr = new_rr
    return x, iters


def trpo_case(dtype, D=5, H=8, A=2, n=24, seed=7):
    g = torch.Generator().manual_seed(seed)
    P = H * D + H + A * H + A
    params0 = (torch.rand(P, generator=g, dtype=torch.float64) * 0.8 - 0.4).to(dtype)
    x = (torch.randn(n, D, generator=g, dtype=torch.float64) * 0.5).to(dtype)
    actions = torch.randint(0, A, (n,), generator=g)
    adv = torch.randn(n, generator=g, dtype=torch.float64).to(dtype)
    v = torch.randn(P, generator=g, dtype=torch.float64).to(dtype)
    dims = (D, H, A)
    params = params0.clone().requires_grad_(True)
    with torch.no_grad():
        lp0 = torch.log_softmax(mlp_forward(params, x, *dims), -1)
        lp0_a = lp0.gather(-1, actions.unsqueeze(-1)).squeeze(-1)
        entropy = categorical_entropy(lp0).mean()
    loss, dist = loss_distance(params, x, actions, adv, lp0, lp0_a, dims)
    (grad,) = torch.autograd.grad(loss, params, retain_graph=True)
    (dgrad,) = torch.autograd.grad(dist, params, create_graph=True)
    reg = 1e-5

    def hvp(vec):
        (hv,) = torch.autograd.grad((dgrad * vec).sum(), params, retain_graph=True)
        return hv + reg * vec

    hv = hvp(v)
    step_dir, iters = solve_cg(hvp, grad, 10, 1e-10)
    step_dir = torch.nan_to_num(step_dir, nan=0.0)
    max_kl = 0.01
    step_size = math.sqrt(1.0 / (float(step_dir.dot(hvp(step_dir))) + 1e-8) * max_kl * 2.0)
    descent = step_size * step_dir
    initial_loss = float(loss)
    prev = params0.clone()
    new_params = prev.clone()
    num_backtracks = -1
    final_loss, final_kl = initial_loss, float("inf")
    for i in range(15):
        ratio = 0.8 ** i
        cand = prev - ratio * descent
        with torch.no_grad():
            l, d = loss_distance(cand, x, actions, adv, lp0, lp0_a, dims)
        final_loss, final_kl = float(l), float(d)
        if final_loss < initial_loss and final_kl <= max_kl:
            num_backtracks = i
            new_params = cand
            break
    ok = (final_loss < initial_loss) and (final_kl < max_kl)
    if not ok:
        new_params = prev
    # (loss, distance) at a perturbed point for the loss/KL evaluation check
    pert = (params0.double() + 0.05 * torch.randn(P, generator=g, dtype=torch.float64)).to(dtype)
    with torch.no_grad():
        l1, d1 = loss_distance(pert, x, actions, adv, lp0, lp0_a, dims)
    return {
        "dims": [D, H, A], "n": n, "dtype": str(dtype).replace("torch.", ""),
        "params0": params0.double().tolist(), "obs": x.double().tolist(), "actions": actions.tolist(),
        "adv": adv.double().tolist(), "v": v.double().tolist(), "reg": reg,
        "entropy": float(entropy), "loss0": initial_loss, "grad": grad.double().tolist(),
        "hvp_v": hv.detach().double().tolist(), "cg_iterations": iters,
        "step_dir": step_dir.detach().double().tolist(), "step_size": step_size,
        "num_backtracks": num_backtracks, "loss_final": final_loss, "kl_final": final_kl, "status_ok": ok,
        "new_params": new_params.detach().double().tolist(),
        "params_pert": pert.double().tolist(), "loss_pert": float(l1), "kl_pert": float(d1),
    }


def critic_case(dtype, D=5, H=8, n=24, steps=6, seed=11):
    g = torch.Generator().manual_seed(seed)
    P = H * D + H + H + 1
    params0 = (torch.rand(P, generator=g, dtype=torch.float64) * 0.8 - 0.4).to(dtype)
    x = (torch.randn(n, D, generator=g, dtype=torch.float64) * 0.5).to(dtype)
    targets = (torch.randn(n, generator=g, dtype=torch.float64) * 3.0).to(dtype)
    params = params0.clone().requires_grad_(True)
    opt = torch.optim.Adam([params], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    losses, grads0 = [], None
    for k in range(steps):
        loss = torch.nn.functional.mse_loss(mlp_forward(params, x, D, H, 1).squeeze(-1), targets)
        opt.zero_grad()
        loss.backward()
        if k == 0:
            grads0 = params.grad.detach().clone()
        losses.append(float(loss))
        opt.step()
    return {
        "dims": [D, H, 1], "n": n, "dtype": str(dtype).replace("torch.", ""), "steps": steps,
        "params0": params0.double().tolist(), "obs": x.double().tolist(), "targets": targets.double().tolist(),
        "grad0": grads0.double().tolist(), "losses": losses, "params_final": params.detach().double().tolist(),
    }


def gae_case(seed=5):
    """GAE / reward-to-go on the reference's own 4-episode history fixture (features.rs:293-333) with a fixed
    random critic on the 1-d boolean observation, computed the reference way on padded sequences."""
    g = torch.Generator().manual_seed(seed)
    D, H = 1, 4
    P = H * D + H + H + 1
    params = (torch.rand(P, generator=g, dtype=torch.float64) * 2 - 1).float()
    episodes = [  # (obs, reward) per step, terminal kind, interrupt successor obs
        ([(1.0, 1.0), (1.0, 1.0), (1.0, 1.0), (1.0, 1.0)], "interrupt_dropped", None),
        ([(0.0, -1.0), (0.0, -1.0), (0.0, 0.0), (0.0, 0.0), (0.0, 1.0), (0.0, 1.0)], "terminate", None),
        ([(0.0, 2.0), (1.0, 2.0), (0.0, 2.0)], "interrupt", 1.0),
        ([(1.0, 3.0)], "terminate", None),
    ]
    gamma, lam = 0.9, 0.8
    out = []
    for steps, kind, succ in episodes:
        obs = torch.tensor([[s[0]] for s in steps], dtype=torch.float32)
        r = torch.tensor([s[1] for s in steps], dtype=torch.float32)
        v = mlp_forward(params, obs, D, H, 1).squeeze(-1)
        if kind == "interrupt":
            vlast = mlp_forward(params, torch.tensor([[succ]], dtype=torch.float32), D, H, 1).squeeze(-1)
        else:
            vlast = torch.zeros(1)
        vnext = torch.cat([v[1:], vlast])
        delta = r + torch.tensor(gamma, dtype=torch.float32) * vnext - v
        adv = delta.clone()
        rtg = r.clone()
        disc = torch.tensor(lam, dtype=torch.float32) * torch.tensor(gamma, dtype=torch.float32)
        for t in range(len(steps) - 2, -1, -1):
            adv[t] = adv[t] + adv[t + 1] * disc
            rtg[t] = rtg[t] + rtg[t + 1] * torch.tensor(gamma, dtype=torch.float32)
        # StepValueTarget::OneStepTd (critics/mod.rs:139-150): rewards + discount_factor * estimated_next_values
        td = r + torch.tensor(gamma, dtype=torch.float32) * vnext
        out.append({"adv": adv.tolist(), "rtg": rtg.tolist(), "values": v.tolist(), "td": td.tolist()})
    return {"critic_params": params.double().tolist(), "dims": [D, H, 1], "gamma": gamma, "lambda": lam,
            "episodes": out}


def dqn_case(dtype, D=5, H=8, A=2, n=30, steps=5, seed=23):
    """DQN loss (dqn.rs:316-326): mse_loss(Q(obs).gather(-1, a).squeeze(-1), targets) + Adam steps, and the
    OneStepTd target r + gamma * amax(Q(next)) with terminal successors masked to 0 (critics/mod.rs:116-148)."""
    g = torch.Generator().manual_seed(seed)
    P = H * D + H + A * H + A
    params0 = (torch.rand(P, generator=g, dtype=torch.float64) * 0.8 - 0.4).to(dtype)
    x = (torch.randn(n, D, generator=g, dtype=torch.float64) * 0.5).to(dtype)
    nx = (torch.randn(n, D, generator=g, dtype=torch.float64) * 0.5).to(dtype)
    actions = torch.randint(0, A, (n,), generator=g)
    rewards = torch.randn(n, generator=g, dtype=torch.float64).to(dtype)
    terminal = torch.rand(n, generator=g) < 0.3
    gamma = 0.97
    with torch.no_grad():
        vnext = mlp_forward(params0, nx, D, H, A).amax(-1)
        vnext = vnext.masked_fill(terminal, 0.0)
        targets = rewards + torch.tensor(gamma, dtype=dtype) * vnext
    params = params0.clone().requires_grad_(True)
    opt = torch.optim.Adam([params], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    losses, grads0 = [], None
    for k in range(steps):
        q = mlp_forward(params, x, D, H, A).gather(-1, actions.unsqueeze(-1)).squeeze(-1)
        loss = torch.nn.functional.mse_loss(q, targets)
        opt.zero_grad()
        loss.backward()
        if k == 0:
            grads0 = params.grad.detach().clone()
        losses.append(float(loss))
        opt.step()
    return {
        "dims": [D, H, A], "n": n, "dtype": str(dtype).replace("torch.", ""), "steps": steps, "gamma": gamma,
        "params0": params0.double().tolist(), "obs": x.double().tolist(), "next_obs": nx.double().tolist(),
        "actions": actions.tolist(), "rewards": rewards.double().tolist(), "terminal": terminal.int().tolist(),
        "td_targets": targets.double().tolist(), "grad0": grads0.double().tolist(), "losses": losses,
        "params_final": params.detach().double().tolist(),
    }


def ppo_case(dtype, D=5, H=8, A=2, n=40, steps=4, seed=31, clip=0.2):
    """Ppo::update (policies/ppo.rs:97-146) and Reinforce::update's loss (reinforce.rs:71-78) with real autograd.
    The parameters are perturbed away from the ones that produced the initial log-probs so that ratios fall on
    both sides of the clip range."""
    g = torch.Generator().manual_seed(seed)
    P = H * D + H + A * H + A
    params_old = (torch.rand(P, generator=g, dtype=torch.float64) * 2.0 - 1.0).to(dtype)
    params0 = (params_old.double() + (torch.rand(P, generator=g, dtype=torch.float64) - 0.5) * 0.6).to(dtype)
    x = (torch.randn(n, D, generator=g, dtype=torch.float64)).to(dtype)
    actions = torch.randint(0, A, (n,), generator=g)
    adv = torch.randn(n, generator=g, dtype=torch.float64).to(dtype)
    adv[0] = 0.0  # a zero advantage: both branches of minimum() tie
    with torch.no_grad():
        lp_old = torch.log_softmax(mlp_forward(params_old, x, D, H, A), -1).gather(-1, actions.unsqueeze(-1)).squeeze(-1)
    params = params0.clone().requires_grad_(True)
    opt = torch.optim.Adam([params], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    losses, grad0, ratios0 = [], None, None
    for k in range(steps):
        lp = torch.log_softmax(mlp_forward(params, x, D, H, A), -1).gather(-1, actions.unsqueeze(-1)).squeeze(-1)
        ratio = (lp - lp_old).exp()
        clipped = ratio.clip(1.0 - clip, 1.0 + clip)
        loss = -torch.minimum(ratio * adv, clipped * adv).mean()
        opt.zero_grad()
        loss.backward()
        if k == 0:
            grad0 = params.grad.detach().clone()
            ratios0 = ratio.detach().clone()
        losses.append(float(loss.detach()))
        opt.step()
    # REINFORCE at params0
    pr = params0.clone().requires_grad_(True)
    lsm = torch.log_softmax(mlp_forward(pr, x, D, H, A), -1)
    lp = lsm.gather(-1, actions.unsqueeze(-1)).squeeze(-1)
    rloss = -(lp * adv).mean()
    rloss.backward()
    entropy = -(lsm.detach() * lsm.detach().exp()).sum(-1).mean()
    return {
        "dims": [D, H, A], "n": n, "dtype": str(dtype).replace("torch.", ""), "steps": steps, "clip": clip,
        "params_old": params_old.double().tolist(), "params0": params0.double().tolist(), "obs": x.double().tolist(),
        "actions": actions.tolist(), "adv": adv.double().tolist(), "logp_old": lp_old.double().tolist(),
        "ratios0": ratios0.double().tolist(), "grad0": grad0.double().tolist(), "losses": losses,
        "params_final": params.detach().double().tolist(),
        "reinforce_loss": float(rloss.detach()), "reinforce_grad": pr.grad.double().tolist(),
        "entropy0": float(entropy),
    }


def gru_case(dtype, D=3, H=4, H2=5, A=2, n=3, T=7, seed=41):
    """Chain<Gru, Mlp> (modules/chain.rs:127-186) over lane trajectories with episode boundaries: per-step outputs
    (torch.gru_cell, relu, two Linear layers), successor outputs at cut episodes, and the gradient of
    sum(dout * out) through time by autograd.  Flat parameter order = trainable_variables()."""
    g = torch.Generator().manual_seed(seed)
    sizes = [3 * H * D, 3 * H * H, 3 * H, 3 * H, H2 * H, H2, A * H2, A]
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
    o = 0
    parts = []
    for sz in sizes:
        parts.append(params[o:o + sz])
        o += sz
    Wih, Whh, bih, bhh = parts[0].reshape(3 * H, D), parts[1].reshape(3 * H, H), parts[2], parts[3]
    W1, b1, W2, b2 = parts[4].reshape(H2, H), parts[5], parts[6].reshape(A, H2), parts[7]

    def head(h):
        return torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(torch.relu(h), W1, b1)), W2, b2)

    out = torch.zeros(A, T, n, dtype=dtype)
    succ = torch.zeros(A, T, n, dtype=dtype)
    outs = []
    for i in range(n):
        h = torch.zeros(1, H, dtype=dtype)
        for t in range(T):
            h = torch.gru_cell(obs[:, t, i].unsqueeze(0), h, Wih, Whh, bih, bhh)
            y = head(h).squeeze(0)
            outs.append((i, t, y))
            f = int(flag[t, i])
            if f == 2 or (f == 0 and t == T - 1):
                xs = term_obs[:, t, i] if f == 2 else obs[:, T, i]
                hs = torch.gru_cell(xs.unsqueeze(0), h, Wih, Whh, bih, bhh)
                succ[:, t, i] = head(hs).squeeze(0).detach()
            if f != 0:
                h = torch.zeros(1, H, dtype=dtype)
    loss = 0
    for i, t, y in outs:
        out[:, t, i] = y.detach()
        loss = loss + (dout[:, t, i] * y).sum()
    loss.backward()
    # forward-mode derivative of all outputs along a random parameter tangent (torch.autograd.functional.jvp)
    tangent = (torch.randn(P, generator=g, dtype=torch.float64)).to(dtype)

    def all_outputs(pv):
        o2 = 0
        pp = []
        for sz in sizes:
            pp.append(pv[o2:o2 + sz])
            o2 += sz
        wih, whh, b_ih, b_hh = pp[0].reshape(3 * H, D), pp[1].reshape(3 * H, H), pp[2], pp[3]
        w1, bb1, w2, bb2 = pp[4].reshape(H2, H), pp[5], pp[6].reshape(A, H2), pp[7]
        ys = []
        for i in range(n):
            h = torch.zeros(1, H, dtype=dtype)
            for t in range(T):
                h = torch.gru_cell(obs[:, t, i].unsqueeze(0), h, wih, whh, b_ih, b_hh)
                y = torch.nn.functional.linear(
                    torch.relu(torch.nn.functional.linear(torch.relu(h), w1, bb1)), w2, bb2).squeeze(0)
                ys.append(y)
                if int(flag[t, i]) != 0:
                    h = torch.zeros(1, H, dtype=dtype)
        return torch.stack(ys)  # [n * T, A]

    _, jv = torch.autograd.functional.jvp(all_outputs, params.detach(), tangent)
    out_dot = jv.reshape(n, T, A).permute(2, 1, 0).contiguous()
    return {
        "tangent": tangent.double().tolist(), "out_dot": out_dot.double().flatten().tolist(),
        "dims": [D, H, H2, A], "n": n, "T": T, "dtype": str(dtype).replace("torch.", ""),
        "params": params.detach().double().tolist(), "obs": obs.double().flatten().tolist(),
        "term_obs": term_obs.double().flatten().tolist(), "flag": flag.flatten().tolist(),
        "dout": dout.double().flatten().tolist(), "out": out.double().flatten().tolist(),
        "succ_out": succ.double().flatten().tolist(), "grad": params.grad.double().tolist(),
    }


def lstm_case(dtype, D=3, H=4, H2=5, A=2, n=3, T=7, seed=43):
    """Chain<Lstm, Mlp> (modules/chain.rs:127-186, seq/rnn/lstm.rs:17-51) over lane trajectories with episode
    boundaries: per-step outputs (torch.lstm_cell with state (h, c), relu, two Linear layers), successor outputs at
    cut episodes, and the gradient of sum(dout * out) through time by autograd.  Gate rows [i; f; g; o]."""
    g = torch.Generator().manual_seed(seed)
    sizes = [4 * H * D, 4 * H * H, 4 * H, 4 * H, H2 * H, H2, A * H2, A]
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
    o = 0
    parts = []
    for sz in sizes:
        parts.append(params[o:o + sz])
        o += sz
    Wih, Whh, bih, bhh = parts[0].reshape(4 * H, D), parts[1].reshape(4 * H, H), parts[2], parts[3]
    W1, b1, W2, b2 = parts[4].reshape(H2, H), parts[5], parts[6].reshape(A, H2), parts[7]

    def head(h):
        return torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(torch.relu(h), W1, b1)), W2, b2)

    out = torch.zeros(A, T, n, dtype=dtype)
    succ = torch.zeros(A, T, n, dtype=dtype)
    outs = []
    zeros = lambda: (torch.zeros(1, H, dtype=dtype), torch.zeros(1, H, dtype=dtype))
    for i in range(n):
        h, c = zeros()
        for t in range(T):
            h, c = torch.lstm_cell(obs[:, t, i].unsqueeze(0), (h, c), Wih, Whh, bih, bhh)
            y = head(h).squeeze(0)
            outs.append((i, t, y))
            f = int(flag[t, i])
            if f == 2 or (f == 0 and t == T - 1):
                xs = term_obs[:, t, i] if f == 2 else obs[:, T, i]
                hs, _ = torch.lstm_cell(xs.unsqueeze(0), (h, c), Wih, Whh, bih, bhh)
                succ[:, t, i] = head(hs).squeeze(0).detach()
            if f != 0:
                h, c = zeros()
    loss = 0
    for i, t, y in outs:
        out[:, t, i] = y.detach()
        loss = loss + (dout[:, t, i] * y).sum()
    loss.backward()
    # forward-mode derivative of all outputs along a random parameter tangent (torch.autograd.functional.jvp)
    tangent = (torch.randn(P, generator=g, dtype=torch.float64)).to(dtype)

    def all_outputs(pv):
        o2 = 0
        pp = []
        for sz in sizes:
            pp.append(pv[o2:o2 + sz])
            o2 += sz
        wih, whh, b_ih, b_hh = pp[0].reshape(4 * H, D), pp[1].reshape(4 * H, H), pp[2], pp[3]
        w1, bb1, w2, bb2 = pp[4].reshape(H2, H), pp[5], pp[6].reshape(A, H2), pp[7]
        ys = []
        for i in range(n):
            h, c = zeros()
            for t in range(T):
                h, c = torch.lstm_cell(obs[:, t, i].unsqueeze(0), (h, c), wih, whh, b_ih, b_hh)
                y = torch.nn.functional.linear(
                    torch.relu(torch.nn.functional.linear(torch.relu(h), w1, bb1)), w2, bb2).squeeze(0)
                ys.append(y)
                if int(flag[t, i]) != 0:
                    h, c = zeros()
        return torch.stack(ys)  # [n * T, A]

    _, jv = torch.autograd.functional.jvp(all_outputs, params.detach(), tangent)
    out_dot = jv.reshape(n, T, A).permute(2, 1, 0).contiguous()
    return {
        "tangent": tangent.double().tolist(), "out_dot": out_dot.double().flatten().tolist(),
        "dims": [D, H, H2, A], "n": n, "T": T, "dtype": str(dtype).replace("torch.", ""),
        "params": params.detach().double().tolist(), "obs": obs.double().flatten().tolist(),
        "term_obs": term_obs.double().flatten().tolist(), "flag": flag.flatten().tolist(),
        "dout": dout.double().flatten().tolist(), "out": out.double().flatten().tolist(),
        "succ_out": succ.double().flatten().tolist(), "grad": params.grad.double().tolist(),
    }


def main():
    data = {
        "generator": "tests/golden/make_torch_golden.py, torch %s CPU" % torch.__version__,
        "trpo_f64": trpo_case(torch.float64),
        "trpo_f32": trpo_case(torch.float32),
        "critic_f32": critic_case(torch.float32),
        "critic_f64": critic_case(torch.float64),
        "gae_f32": gae_case(),
    }
    with open(os.path.join(HERE, "torch_golden.json"), "w") as f:
        json.dump(data, f)
    print("wrote torch_golden.json")
    dqn = {
        "generator": "tests/golden/make_torch_golden.py, torch %s CPU" % torch.__version__,
        "dqn_f32": dqn_case(torch.float32),
        "dqn_f64": dqn_case(torch.float64),
    }
    with open(os.path.join(HERE, "torch_golden_dqn.json"), "w") as f:
        json.dump(dqn, f)
    print("wrote torch_golden_dqn.json")
    ppo = {
        "generator": "tests/golden/make_torch_golden.py, torch %s CPU" % torch.__version__,
        "ppo_f32": ppo_case(torch.float32),
        "ppo_f64": ppo_case(torch.float64),
    }
    with open(os.path.join(HERE, "torch_golden_ppo.json"), "w") as f:
        json.dump(ppo, f)
    print("wrote torch_golden_ppo.json")
    gru = {
        "generator": "tests/golden/make_torch_golden.py, torch %s CPU" % torch.__version__,
        "gru_f32": gru_case(torch.float32),
        "gru_f64": gru_case(torch.float64),
    }
    with open(os.path.join(HERE, "torch_golden_gru.json"), "w") as f:
        json.dump(gru, f)
    print("wrote torch_golden_gru.json")
    lstm = {
        "generator": "tests/golden/make_torch_golden.py, torch %s CPU" % torch.__version__,
        "lstm_f32": lstm_case(torch.float32),
        "lstm_f64": lstm_case(torch.float64),
    }
    with open(os.path.join(HERE, "torch_golden_lstm.json"), "w") as f:
        json.dump(lstm, f)
    print("wrote torch_golden_lstm.json")


if __name__ == "__main__":
    main()
