"""Synthetic batches generated ON the GPU with the engine's own sign / key-derivation kernels
(the reference shape: sk, message, nonce -> sig, pk; benches/signature.rs:48-60), then corrupted
in a known pattern so the expected verdict vector is non-trivial.

Scalars are drawn as 251-bit (sk, nonce) / 254-bit (message) uniform integers: canonical by
construction (2^251 < r, 2^254 < q).  That is not exactly `Fr::random` (64 bytes mod r) but the
verification cost does not depend on the distribution.
"""
import torch

from . import engine as E

TAMPER_PERIOD = 16


def _rand_scalars(n, top_mask, gen, device):
    x = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=device, generator=gen)
    x[:, 31] &= top_mask
    return x.contiguous()


def _tamper(batch, n):
    """Corrupt items i = 0 (mod 16), cycling 4 classes; returns expected verdicts."""
    dev = batch["u"].device
    expected = torch.ones(n, dtype=torch.uint8, device=dev)
    idx = torch.arange(0, n, TAMPER_PERIOD, device=dev)
    cls = (idx // TAMPER_PERIOD) % 4
    expected[idx] = 0
    i0 = idx[cls == 0]
    batch["PK"][i0] = batch["PK"][(i0 + 1) % n]          # wrong key
    i1 = idx[cls == 1]
    batch["u"][i1, 0] ^= 8                                # bit flip in u
    i2 = idx[cls == 2]
    batch["m"][i2, 9] ^= 32                               # bit flip in the message
    i3 = idx[cls == 3]
    batch["R"][i3] = batch["R"][(i3 + 2) % n]             # another signature's R
    return expected


def gen_single(n, seed, device="cuda:0", tamper=True):
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    sk = _rand_scalars(n, 0x07, gen, device)
    r = _rand_scalars(n, 0x07, gen, device)
    m = _rand_scalars(n, 0x3F, gen, device)
    u = torch.empty((n, 32), dtype=torch.uint8, device=device)
    R = torch.empty((n, 64), dtype=torch.uint8, device=device)
    PK = torch.empty((n, 64), dtype=torch.uint8, device=device)
    E.sign_single_dev(sk, m, r, u, R)
    E.public_keys_dev(sk, 0, PK)
    torch.cuda.synchronize()
    batch = {"u": u, "R": R, "PK": PK, "m": m}
    batch["expected"] = _tamper(batch, n) if tamper else torch.ones(n, dtype=torch.uint8, device=device)
    return batch


def gen_double(n, seed, device="cuda:0", tamper=True):
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    sk = _rand_scalars(n, 0x07, gen, device)
    r = _rand_scalars(n, 0x07, gen, device)
    m = _rand_scalars(n, 0x3F, gen, device)
    u = torch.empty((n, 32), dtype=torch.uint8, device=device)
    R = torch.empty((n, 64), dtype=torch.uint8, device=device)
    Rp = torch.empty((n, 64), dtype=torch.uint8, device=device)
    PK = torch.empty((n, 64), dtype=torch.uint8, device=device)
    PKp = torch.empty((n, 64), dtype=torch.uint8, device=device)
    E.sign_double_dev(sk, m, r, u, R, Rp)
    E.public_keys_dev(sk, 0, PK)
    E.public_keys_dev(sk, 1, PKp)
    torch.cuda.synchronize()
    batch = {"u": u, "R": R, "Rp": Rp, "PK": PK, "PKp": PKp, "m": m}
    if tamper:
        expected = _tamper(batch, n)
        # additionally break ONLY the primed half of items i = 8 (mod 16)
        idx = torch.arange(8, n, TAMPER_PERIOD, device=u.device)
        batch["PKp"][idx] = batch["PKp"][(idx + 1) % n]
        expected[idx] = 0
        batch["expected"] = expected
    else:
        batch["expected"] = torch.ones(n, dtype=torch.uint8, device=device)
    return batch


def gen_vargen(n, seed, device="cuda:0", tamper=True):
    """Var-generator scheme (benches/signature_var_generator.rs:50-63 shape): sk, generator scalar
    g, message, nonce; Gen = g*G, PK = sk*Gen, R = r*Gen.  Generated through the host entry points
    (the variable-base sign/derive kernels have no device-pointer form), then moved to HBM."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    sk = _rand_scalars(n, 0x07, gen, device).cpu().numpy()
    g = _rand_scalars(n, 0x07, gen, device).cpu().numpy()
    r = _rand_scalars(n, 0x07, gen, device).cpu().numpy()
    m = _rand_scalars(n, 0x3F, gen, device).cpu().numpy()
    Gen = E.public_keys(g, 0)
    PK = E.public_keys(sk, 0, Gen)
    u, R = E.sign_vargen(sk, Gen, m, r)
    to = lambda a: torch.from_numpy(a).to(device).contiguous()
    batch = {"u": to(u), "R": to(R), "PK": to(PK), "Gen": to(Gen), "m": to(m)}
    if tamper:
        expected = _tamper(batch, n)
        idx = torch.arange(8, n, TAMPER_PERIOD, device=device)
        batch["Gen"][idx] = batch["Gen"][(idx + 1) % n]      # wrong generator
        expected[idx] = 0
        batch["expected"] = expected
    else:
        batch["expected"] = torch.ones(n, dtype=torch.uint8, device=device)
    return batch
