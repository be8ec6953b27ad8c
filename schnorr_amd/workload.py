"""Synthetic batches generated ON the GPU with the engine's own kernels, in the reference
harness's shape (tests/schnorr.rs:16-22, benches/signature.rs:48-60): per item sk, message and
nonce are consecutive draws of `StdRng::seed_from_u64(seed)` (ChaCha12, each `from_bytes_wide` of
64 bytes — dsv_stdrng_sign_inputs), then sig = sign(sk, m, nonce), pk = sk*G.  Every 16th item is
corrupted in a known pattern so the expected verdict vector is non-trivial.
`first_item` selects a slice of the one stream, so N ranks hold consecutive shards of one batch.
The var-generator batches follow the same rule with four draws per item (sk, generator scalar,
message, nonce: src/keys/secret.rs:371-373, tests/schnorr_var_generator.rs:16-22).
"""
import torch

from . import engine as E

TAMPER_PERIOD = 16


def _rand_scalars(n, top_mask, gen, device):
    x = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=device, generator=gen)
    x[:, 31] &= top_mask
    return x.contiguous()


def _arange(lo, hi, step, device):
    """torch.arange that tolerates an empty range (tiny batches)"""
    if hi <= lo:
        return torch.empty(0, dtype=torch.int64, device=device)
    return torch.arange(lo, hi, step, device=device)


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


def _harness_inputs(n, seed, first_item, device):
    sk = torch.empty((n, 32), dtype=torch.uint8, device=device)
    m = torch.empty((n, 32), dtype=torch.uint8, device=device)
    r = torch.empty((n, 32), dtype=torch.uint8, device=device)
    E.stdrng_sign_inputs_dev(seed, sk, m, r, first_item=first_item)
    return sk, m, r


def gen_single(n, seed, device="cuda:0", tamper=True, first_item=0):
    sk, m, r = _harness_inputs(n, seed, first_item, device)
    u = torch.empty((n, 32), dtype=torch.uint8, device=device)
    R = torch.empty((n, 64), dtype=torch.uint8, device=device)
    PK = torch.empty((n, 64), dtype=torch.uint8, device=device)
    E.sign_single_dev(sk, m, r, u, R)
    E.public_keys_dev(sk, 0, PK)
    torch.cuda.synchronize()
    batch = {"u": u, "R": R, "PK": PK, "m": m}
    batch["expected"] = _tamper(batch, n) if tamper else torch.ones(n, dtype=torch.uint8, device=device)
    return batch


def gen_double(n, seed, device="cuda:0", tamper=True, first_item=0):
    sk, m, r = _harness_inputs(n, seed, first_item, device)
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
        idx = _arange(8, n, TAMPER_PERIOD, u.device)
        batch["PKp"][idx] = batch["PKp"][(idx + 1) % n]
        expected[idx] = 0
        batch["expected"] = expected
    else:
        batch["expected"] = torch.ones(n, dtype=torch.uint8, device=device)
    return batch


def gen_vargen(n, seed, device="cuda:0", tamper=True, first_item=0):
    """Var-generator scheme (benches/signature_var_generator.rs:50-63 shape): per item sk, generator
    scalar g, message and nonce are consecutive `from_bytes_wide` draws of StdRng(seed);
    Gen = g*G, PK = sk*Gen, R = r*Gen — all computed on the GPU."""
    new = lambda w: torch.empty((n, w), dtype=torch.uint8, device=device)
    sk, g, m, r = new(32), new(32), new(32), new(32)
    E.stdrng_vargen_inputs_dev(seed, sk, g, m, r, first_item=first_item)
    Gen, PK, u, R = new(64), new(64), new(32), new(64)
    ws = torch.empty(E.workspace_bytes(n), dtype=torch.uint8, device=device)
    E.public_keys_dev(g, 0, Gen)
    E.public_keys_vargen_dev(sk, Gen, PK, ws)
    E.sign_vargen_dev(sk, Gen, m, r, u, R, ws)
    torch.cuda.synchronize(device)
    batch = {"u": u, "R": R, "PK": PK, "Gen": Gen, "m": m}
    if tamper:
        expected = _tamper(batch, n)
        idx = _arange(8, n, TAMPER_PERIOD, device)
        batch["Gen"][idx] = batch["Gen"][(idx + 1) % n]      # wrong generator
        expected[idx] = 0
        batch["expected"] = expected
    else:
        batch["expected"] = torch.ones(n, dtype=torch.uint8, device=device)
    return batch


def gen_mixed(n, seed, device="cuda:0", tamper=True, first_item=0):
    """BASELINE.json configs[4] shape: n items, single (kind 0) on even and double (kind 1) on odd
    positions, as ONE structure of arrays (Rp / PKp rows of single items are zero).  The n/2
    singles are items first_item/2.. of StdRng(seed), the doubles the same range of
    StdRng(seed + 1), so consecutive `first_item` ranges of N ranks tile one global batch."""
    assert n % 2 == 0 and first_item % 2 == 0
    h = n // 2
    bs = gen_single(h, seed, device, tamper, first_item // 2)
    bd = gen_double(h, seed + 1, device, tamper, first_item // 2)
    out = {"kinds": (torch.arange(n, device=device) & 1).to(torch.uint8)}
    for key, w in (("u", 32), ("R", 64), ("Rp", 64), ("PK", 64), ("PKp", 64), ("m", 32)):
        t = torch.zeros((n, w), dtype=torch.uint8, device=device)
        if key in bs:
            t[0::2] = bs[key]
        t[1::2] = bd[key]
        out[key] = t.contiguous()
    exp = torch.empty(n, dtype=torch.uint8, device=device)
    exp[0::2] = bs["expected"]
    exp[1::2] = bd["expected"]
    out["expected"] = exp
    out["n_double"] = h
    return out
