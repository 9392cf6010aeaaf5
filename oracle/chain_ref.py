"""CPU oracle for the LF-MMI ("chain") objective -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product path (pykaldi2_amd/) never does.

What it restates
----------------
The reference computes the LF-MMI objective by calling
``kaldi.chain.compute_chain_objf_and_deriv`` (reference ops/ops.py:265, wrapped
by ``ChainObjtiveFunction`` ops/ops.py:243-280; graph built at
bin/train_chain.py:167,202).  Kaldi / PyKaldi are third-party, un-vendored and
un-pinned (reference docker/Dockerfile:57-64 clones pykaldi@master) and are
absent from /root/reference, so this file restates Kaldi's *published*
algorithm (chain-den-graph.cc, chain-denominator.cc, chain-numerator.cc,
chain-training.cc; summarised in SURVEY.md Appendix A).

PARITY UNPINNED at the Kaldi boundary: the reference holds no golden vectors,
tests or fixtures for this path.  The pins this
oracle carries instead are self-made and run in tests/test_oracle_chain.py:
  (i)   brute-force path enumeration on tiny graphs (``brute_force_den``),
  (ii)  a float64 log-domain PyTorch autograd restatement (``torch_logdomain_den``),
  (iii) invariants: per-frame occupancies sum to 1, grad sums to 0 per frame.

Everything here is plain numpy float64 unless a dtype is passed.
"""
import numpy as np


# --------------------------------------------------------------------------
# Denominator graph (Kaldi DenominatorGraph)
# --------------------------------------------------------------------------
class DenGraphRef:
    """Arc-list view of den.fst: arc a = (src[a] -> dst[a], pdf[a], prob[a]).

    prob = exp(-weight), pdf = ilabel - 1; final weights are ignored
    (chain-den-graph.cc SetTransitions).  ``initial_probs`` follows
    DenominatorGraph::SetInitialProbs: 100 iterations of the normalised
    forward recursion from the start state, averaged.
    """

    def __init__(self, num_states, src, dst, pdf, prob, start=0, num_pdfs=None):
        self.S = int(num_states)
        self.src = np.asarray(src, dtype=np.int64)
        self.dst = np.asarray(dst, dtype=np.int64)
        self.pdf = np.asarray(pdf, dtype=np.int64)
        self.prob = np.asarray(prob, dtype=np.float64)
        self.start = int(start)
        self.P = int(num_pdfs if num_pdfs is not None else self.pdf.max() + 1)
        self.initial_probs = initial_probs_ref(self.S, self.src, self.dst, self.prob, self.start)


def initial_probs_ref(S, src, dst, prob, start, iters=100):
    cur = np.zeros(S)
    cur[start] = 1.0
    avg = np.zeros(S)
    for _ in range(iters):
        avg += cur / iters
        nxt = np.bincount(dst, weights=cur[src] * prob, minlength=S)
        cur = nxt / nxt.sum()
    return avg


BETA_SEED = 1.0     # test hook: beta'[T] = BETA_SEED / tot (tests of the abandon rule on real inputs); 1.0 = Kaldi


def den_forward_backward(logits, g, leaky, dtype=np.float64):
    """Denominator forward-backward, probability space with per-frame rescaling.

    Follows SURVEY Appendix A.2 line by line (Kaldi DenominatorComputation::
    Forward/Backward, AlphaFirstFrame / AlphaGeneralFrame / AlphaDash /
    BetaDashLastFrame / BetaDashGeneralFrame / Beta):

      x[t,p]   = exp(clamp(logit[t,p], -30, 30))
      alpha[0] = pi ; asum[t] = sum alpha[t] ; alpha'[t] = alpha[t] + l*asum[t]*pi
      alpha[t+1,d] = sum_{s->d} alpha'[t,s] * prob * x[t,pdf] / asum[t]
      log p_den = log(sum alpha'[T]) + sum_{t<T} log asum[t]
      beta'[T] = 1/tot ; beta = beta' + l * sum_k pi[k] beta'[k]
      gamma[t,pdf] += alpha'[t,s] * prob * x[t,pdf] * beta[t+1,d] / asum[t]

    Returns (logprob, gamma[T,P]) with gamma = d logprob / d logit.
    """
    logits = np.asarray(logits)
    T, P = logits.shape
    S = g.S
    pi = g.initial_probs.astype(dtype)
    prob = g.prob.astype(dtype)
    x = np.exp(np.clip(logits, -30.0, 30.0)).astype(dtype)
    ell = dtype(leaky)

    alpha_dash = np.zeros((T + 1, S), dtype=dtype)
    asum = np.zeros(T + 1, dtype=dtype)
    a = pi.copy()
    for t in range(T + 1):
        asum[t] = a.sum()
        alpha_dash[t] = a + ell * asum[t] * pi
        if t == T:
            break
        contrib = alpha_dash[t][g.src] * prob * x[t][g.pdf]
        a = (np.bincount(g.dst, weights=contrib, minlength=S) / asum[t]).astype(dtype)
    tot = alpha_dash[T].sum()
    logprob = np.log(np.float64(tot)) + np.log(asum[:T].astype(np.float64)).sum()

    gamma = np.zeros((T, P), dtype=dtype)
    beta_dash = np.full(S, BETA_SEED / tot, dtype=dtype)
    beta = beta_dash + ell * (pi * beta_dash).sum()
    for t in range(T - 1, -1, -1):
        f = prob * x[t][g.pdf] * beta[g.dst] / asum[t]
        gamma[t] = np.bincount(g.pdf, weights=alpha_dash[t][g.src] * f, minlength=P)
        beta_dash = np.bincount(g.src, weights=f, minlength=S).astype(dtype)
        beta = beta_dash + ell * (pi * beta_dash).sum()
    # Kaldi's consistency check (BetaGeneralFrameDebug): sum_h alpha'[0,h] * beta'[0,h] == 1
    check = float((alpha_dash[0] * beta_dash).sum())
    return float(logprob), gamma, check


def brute_force_den(logits, g, leaky):
    """Exact log p_den by dense matrix products in float64 (tiny graphs only).

    The leaky-HMM is the rank-one extra transition  l * (sum_s alpha[s]) * pi[d]
    applied after every frame *including* t=0 and t=T (Appendix A.2), i.e.
    alpha' = alpha (I + l * 1 pi^T).
    """
    logits = np.asarray(logits, dtype=np.float64)
    T, P = logits.shape
    S = g.S
    x = np.exp(np.clip(logits, -30, 30))
    pi = g.initial_probs
    leak = np.eye(S) + leaky * np.outer(np.ones(S), pi)
    v = pi @ leak
    for t in range(T):
        M = np.zeros((S, S))
        np.add.at(M, (g.src, g.dst), g.prob * x[t][g.pdf])
        v = (v @ M) @ leak
    return float(np.log(v.sum()))


def torch_logdomain_den(logits, g, leaky):
    """float64 log-domain PyTorch restatement, differentiated by autograd
    (SURVEY Appendix A.4 recipe).  Returns (logprob, grad[T,P])."""
    import torch
    lg = torch.tensor(np.asarray(logits, dtype=np.float64), requires_grad=True)
    T, P = lg.shape
    S = g.S
    src = torch.tensor(g.src)
    dst = torch.tensor(g.dst)
    pdf = torch.tensor(g.pdf)
    lprob = torch.log(torch.tensor(g.prob))
    lpi = torch.log(torch.tensor(g.initial_probs).clamp_min(1e-300))
    lleak = np.log(leaky) if leaky > 0 else -np.inf
    xl = lg.clamp(-30.0, 30.0)

    def leak(la):
        tot = torch.logsumexp(la, 0)
        return torch.logaddexp(la, lleak + tot + lpi)

    la = leak(lpi)
    neg_inf = torch.full((S,), -1e300, dtype=torch.float64)
    for t in range(T):
        v = la[src] + lprob + xl[t][pdf]
        m = neg_inf.scatter_reduce(0, dst, v, reduce="amax", include_self=True)
        s = torch.zeros(S, dtype=torch.float64).scatter_add(0, dst, torch.exp(v - m[dst]))
        la = leak(m + torch.log(s.clamp_min(1e-300)))
    lp = torch.logsumexp(la, 0)
    lp.backward()
    return float(lp), lg.grad.numpy()


# --------------------------------------------------------------------------
# Numerator (Kaldi NumeratorComputation)
# --------------------------------------------------------------------------
class NumFstRef:
    """Acyclic, top-sorted supervision FST: every arc consumes one frame.
    arc a: src[a] -> dst[a], label pdf[a] (0-based), weight w[a] (-log prob).
    ``final`` maps state -> final weight (absent = non-final).  ``time[s]`` is
    the frame index of state s (number of frames consumed on any path to s)."""

    def __init__(self, num_states, src, dst, pdf, weight, final_states, final_weights, time):
        self.S = int(num_states)
        self.src = np.asarray(src, dtype=np.int64)
        self.dst = np.asarray(dst, dtype=np.int64)
        self.pdf = np.asarray(pdf, dtype=np.int64)
        self.w = np.asarray(weight, dtype=np.float64)
        self.final_states = np.asarray(final_states, dtype=np.int64)
        self.final_weights = np.asarray(final_weights, dtype=np.float64)
        self.time = np.asarray(time, dtype=np.int64)


def num_forward_backward(logits, f):
    """Log-domain forward-backward over the supervision FST (Appendix A.3):
    log alpha[0] = 0; log alpha[d] (+)= log alpha[s] - w + logit[time(s), pdf];
    log p_num = (+)_{final} (log alpha[s] - final_w).  Returns (logprob,
    post[T,P]) where post rows sum to 1."""
    logits = np.asarray(logits, dtype=np.float64)
    T, P = logits.shape
    NEG = -np.inf
    la = np.full(f.S, NEG)
    la[0] = 0.0
    order = np.argsort(f.time[f.src], kind="stable")
    src, dst, pdf, w = f.src[order], f.dst[order], f.pdf[order], f.w[order]
    tt = f.time[src]
    score = -w + logits[tt, pdf]
    bounds = np.searchsorted(tt, np.arange(T + 1))
    for t in range(T):
        lo, hi = bounds[t], bounds[t + 1]
        v = la[src[lo:hi]] + score[lo:hi]
        np.logaddexp.at(la, dst[lo:hi], v)
    lp = np.logaddexp.reduce(la[f.final_states] - f.final_weights)
    lb = np.full(f.S, NEG)
    lb[f.final_states] = -f.final_weights
    post = np.zeros((T, P))
    for t in range(T - 1, -1, -1):
        lo, hi = bounds[t], bounds[t + 1]
        v = score[lo:hi] + lb[dst[lo:hi]]
        np.logaddexp.at(lb, src[lo:hi], v)
        np.add.at(post[t], pdf[lo:hi], np.exp(la[src[lo:hi]] + v - lp))
    return float(lp), post


# --------------------------------------------------------------------------
# ComputeChainObjfAndDeriv (Appendix A.1) + the reference's own post-processing
# --------------------------------------------------------------------------
def chain_objf_and_deriv(logits, g, fst, leaky=1e-4, xent_regularize=0.0, weight=1.0,
                         l2_regularize=0.0):
    """objf = w*(log p_num - log p_den); grad = w*(num_post - den_post)
    (+ xent_regularize * w * num_post, reference ops/ops.py:267); NaN guard of
    Appendix A.1 step 4.  Returns (objf, grad[T,P], aux dict).  The reference
    hands -grad to autograd (ops/ops.py:276-280)."""
    T = logits.shape[0]
    den_lp, den_post, check = den_forward_backward(logits, g, leaky)
    num_lp, num_post = num_forward_backward(logits, fst)
    objf = weight * (num_lp - den_lp)
    grad = weight * (num_post - den_post)
    grad_xent = weight * num_post
    ok = np.isfinite(objf) and abs(check - 1.0) <= 2.0
    if not ok:
        objf = -10.0 * weight * T
        grad = np.zeros_like(grad)
        grad_xent = np.zeros_like(grad_xent)
    if l2_regularize != 0.0:
        grad = grad - weight * l2_regularize * np.asarray(logits, dtype=np.float64)
    grad = grad + xent_regularize * grad_xent
    return float(objf), grad, dict(num=num_lp, den=den_lp, check=check, ok=ok)
