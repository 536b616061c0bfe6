"""Reads the integer tables of the reference's C sources as TEXT (container only: /root/reference does not exist
on the GPU box, tests using this skip there).  Nothing is copied into the repo: the tables are parsed at test time
and compared with the oracle's and the product's own transcriptions.

  conv.c    every `next_output` / `next_state` array, every `struct osmo_conv_code` initialiser, and the generator
            polynomials written in the comment above each code (reference src/l1/conv.c)
  punct.c   all 51 `struct gmr1_puncturer` initialisers (src/l1/punct.c:136-1166)
  nb.c      the 10 burst formats: sync chunks, data chunks, guard / len / ebits, modulation (src/sdr/nb.c)
"""
from __future__ import annotations

import os
import re

REF = "/root/reference"


def available() -> bool:
    return os.path.isfile(os.path.join(REF, "src", "l1", "conv.c"))


def _read(rel):
    with open(os.path.join(REF, rel)) as f:
        return f.read()


def _strip_comments(s):
    return re.sub(r"/\*.*?\*/", " ", s, flags=re.S)


def _ints(s):
    return [int(x, 0) for x in re.findall(r"-?\b(?:0x[0-9a-fA-F]+|\d+)\b", s)]


def _const_expr(s):
    """`39 * 6` style initialisers."""
    s = s.strip()
    assert re.fullmatch(r"[\d\s*+\-()]+", s), s
    return int(eval(s, {"__builtins__": {}}))


def parse_conv():
    """-> {code name: dict(N, K, next_output [[o0, o1]...], next_state [[s0, s1]...], polys_comment [int...])}.
    polys_comment: the g_i(D) of the comment above the code's output table as masks, bit i = D^i;
    polys_table: the generators the printed table itself implements (read off the unit states)."""
    raw = _read("src/l1/conv.c")
    # comment polynomials: the block comment directly before a `..._next_output[][2]` array
    polys_of = {}
    for m in re.finditer(r"/\*((?:(?!\*/).)*?)\*/\s*static const uint8_t (\w+)_next_output\[\]\[2\]", raw, flags=re.S):
        terms = re.findall(r"g(\d)\(D\)\s*=\s*([^\n]*)", m.group(1))
        polys = []
        for _, rhs in terms:
            # terms by token: a few of the comments leave a `+` out ("D^3         D^5")
            exps = set()
            for t in re.findall(r"D\^\d+|D|1", rhs):
                exps.add(0 if t == "1" else 1 if t == "D" else int(t[2:]))
            polys.append(exps)
        polys_of[m.group(2)] = polys
    src = _strip_comments(raw)
    arrays = {}
    for m in re.finditer(r"static const uint8_t (\w+)\[\]\[2\]\s*=\s*\{(.*?)\};", src, flags=re.S):
        v = _ints(m.group(2))
        assert len(v) % 2 == 0
        arrays[m.group(1)] = [[v[i], v[i + 1]] for i in range(0, len(v), 2)]
    codes = {}
    for m in re.finditer(r"const struct osmo_conv_code (\w+)\s*=\s*\{(.*?)\};", src, flags=re.S):
        body = m.group(2)
        f = dict(re.findall(r"\.(\w+)\s*=\s*([^,]+),", body))
        N, K = int(f["N"]), int(f["K"])
        c = dict(N=N, K=K, next_output=arrays[f["next_output"].strip()], next_state=arrays[f["next_state"].strip()],
                 term=f["term"].strip())
        key = f["next_output"].strip()[:-len("_next_output")]
        exps = polys_of[key]
        assert len(exps) == N, (m.group(1), exps)
        c["polys_comment"] = [sum(1 << e for e in es) for es in exps]
        # a feed-forward code is linear: input 1 from state 0 shows the D^0 taps, state 1 << (i - 1) with input 0 the D^i taps
        taps = [c["next_output"][0][1]] + [c["next_output"][1 << (i - 1)][0] for i in range(1, K)]
        c["polys_table"] = [sum(((taps[i] >> (N - 1 - n)) & 1) << i for i in range(K)) for n in range(N)]
        codes[m.group(1)] = c
    return codes


def trellis_from_polys(N, K, polys):
    """Feed-forward encoder tables from generator masks (bit i = D^i); state = the last K-1 inputs, newest in bit 0."""
    ns = 1 << (K - 1)
    out, nxt = [], []
    for s in range(ns):
        ro, rs = [], []
        for b in (0, 1):
            reg = (s << 1) | b                                 # D^i at bit i
            w = 0
            for g in polys:
                bit = bin(reg & g).count("1") & 1
                w = (w << 1) | bit                             # first generator in the MSB
            ro.append(w)
            rs.append(reg & (ns - 1))
        out.append(ro)
        nxt.append(rs)
    return out, nxt


def parse_punct():
    """-> {name: dict(r, L, N, mask [0/1...])}, in file order."""
    src = _strip_comments(_read("src/l1/punct.c"))
    out = {}
    for m in re.finditer(r"const struct gmr1_puncturer (\w+)\s*=\s*\{(.*?)\n\};", src, flags=re.S):
        body = m.group(2)
        f = {k: int(v) for k, v in re.findall(r"\.(r|L|N)\s*=\s*(\d+)", body)}
        mask = _ints(re.search(r"\.mask\s*=\s*\{(.*?)\}", body, flags=re.S).group(1))
        out[m.group(1)] = dict(r=f["r"], L=f["L"], N=f["N"], mask=mask)
    return out


def puncturer_generate(N, coded_len, pre, main, post, repeat):
    """gmr1_puncturer_generate (src/l1/punct.c:48-133) over parsed masks: ascending punctured positions of the
    unpunctured coded stream of `coded_len` bits."""
    cl = coded_len
    if pre:
        cl -= pre["L"] * N
    if post:
        cl -= post["L"] * N
    d = main["L"] * N
    if not repeat:
        repeat = (cl + d - 1) // d
    p = []
    cl = coded_len
    ii = 0
    if pre:
        for ip in range(pre["L"] * N):
            if ii >= cl:
                break
            if pre["mask"][ip] == 0:
                p.append(ii)
            ii += 1
    if post:
        cl -= post["L"] * N
    for _ in range(repeat):
        for ip in range(main["L"] * N):
            if ii >= cl:
                break
            if main["mask"][ip] == 0:
                p.append(ii)
            ii += 1
    if post:
        ii = cl
        for ip in range(post["L"] * N):
            if post["mask"][ip] == 0:
                p.append(ii)
            ii += 1
    return p


def parse_nb():
    """-> {burst name ('bcch', 'dc2', ...): dict(mod, guard_pre, guard_post, len, ebits, sync [[(pos, [syms])...]...],
    data [(pos, len)...])}."""
    src = _strip_comments(_read("src/sdr/nb.c"))
    sync, data = {}, {}
    for m in re.finditer(r"static struct gmr1_pi4cxpsk_sync (\w+)\[\]\s*=\s*\{(.*?)\n\};", src, flags=re.S):
        chunks = []
        for e in re.finditer(r"\{\s*(-?\d+)\s*(?:,\s*(\d+)\s*,\s*\{([^}]*)\}\s*)?\}", m.group(2)):
            if int(e.group(1)) < 0:
                break
            syms = _ints(e.group(3))
            assert len(syms) == int(e.group(2)), m.group(1)
            chunks.append((int(e.group(1)), syms))
        sync[m.group(1)] = chunks
    for m in re.finditer(r"static struct gmr1_pi4cxpsk_data (\w+)\[\]\s*=\s*\{(.*?)\n\};", src, flags=re.S):
        chunks = []
        for e in re.finditer(r"\{\s*(-?\d+)\s*(?:,\s*(\d+)\s*)?\}", m.group(2)):
            if int(e.group(1)) < 0:
                break
            chunks.append((int(e.group(1)), int(e.group(2))))
        data[m.group(1)] = chunks
    bursts = {}
    for m in re.finditer(r"struct gmr1_pi4cxpsk_burst gmr1_(\w+)_burst\s*=\s*\{(.*?)\n\};", src, flags=re.S):
        body = m.group(2)
        f = dict(re.findall(r"\.(\w+)\s*=\s*(\{[^}]*\}|[^,{]+),", body))
        seqs = [s.strip() for s in f["sync"].strip("{} ").split(",")]
        seqs = [s for s in seqs if s and s != "NULL"]
        bursts[m.group(1)] = dict(mod=f["mod"].strip().lstrip("&"), guard_pre=int(f["guard_pre"]),
                                  guard_post=int(f["guard_post"]), len=_const_expr(f["len"]),
                                  ebits=_const_expr(f["ebits"]), sync=[sync[s] for s in seqs],
                                  data=data[f["data"].strip()])
    return bursts


def parse_modulations():
    """-> {'gmr1_pi4cqpsk': dict(rotation_expr, nbits, syms [(idx, data bits..., re, im)])} from src/sdr/pi4cxpsk.c:71-115."""
    src = _strip_comments(_read("src/sdr/pi4cxpsk.c"))
    mods = {}
    for m in re.finditer(r"struct gmr1_pi4cxpsk_modulation (\w+)\s*=\s*\{(.*?)\n\};", src, flags=re.S):
        body = m.group(2)
        rot = re.search(r"\.rotation\s*=\s*([^,]+),", body).group(1).strip()
        nbits = int(re.search(r"\.nbits\s*=\s*(\d+)", body).group(1))
        mods[m.group(1)] = dict(rotation_expr=rot, nbits=nbits)
    return mods
