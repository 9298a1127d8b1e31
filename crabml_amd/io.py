"""Sequence / tree readers feeding the likelihood boundary.

Host-side mirror of the small part of crabML's I/O layer that the hot path's
callers rely on (reference: src/crabml/io/sequences.py:37-52,305-338 for the
codon encoding, src/crabml/io/trees.py:67-206 for Newick with ``#n`` labels).
Written from scratch around flat numpy arrays -- the engine wants arrays,
not node objects.

Codon state space: the 61 sense codons in PAML order, index = rank among the 64
codons enumerated as n0*16 + n1*4 + n2 with T=0, C=1, A=2, G=3, skipping TAA,
TAG, TGA.  Gap codon ``---`` is code 64, anything else unreadable is -1; both
mean "missing" to the likelihood (any code outside [0, 61)).
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

NUCS = "TCAG"
_AA = ("FFLLSSSSYY**CC*W" "LLLLPPPPHHQQRRRR" "IIIMTTTTNNKKSSRR" "VVVVAAAADDEEGGGG")
CODONS64 = [a + b + c for a in NUCS for b in NUCS for c in NUCS]
GENETIC_CODE = dict(zip(CODONS64, _AA))
CODONS = [c for c in CODONS64 if GENETIC_CODE[c] != "*"]  # 61 sense codons
CODON_TO_INDEX = {c: i for i, c in enumerate(CODONS)}
N_CODONS = 61
GAP_CODE = 64
UNKNOWN_CODE = -1


def encode_codon_string(seq: str) -> np.ndarray:
    """Nucleotide string -> int8 codon codes (0..60, 64 gap, -1 unknown)."""
    seq = seq.upper().replace("U", "T")
    n = len(seq) // 3
    out = np.empty(n, dtype=np.int8)
    for j in range(n):
        c = seq[3 * j:3 * j + 3]
        out[j] = GAP_CODE if c == "---" else CODON_TO_INDEX.get(c, UNKNOWN_CODE)
    return out


@dataclass
class Alignment:
    """Codon alignment: ``sequences[i, s]`` is the code of taxon ``names[i]`` at site s."""

    names: list
    sequences: np.ndarray  # int8 [n_species, n_sites]
    seqtype: str = "codon"

    @property
    def n_species(self) -> int:
        return self.sequences.shape[0]

    @property
    def n_sites(self) -> int:
        return self.sequences.shape[1]

    # -- readers ---------------------------------------------------------
    @classmethod
    def from_phylip(cls, path, seqtype: str = "codon") -> "Alignment":
        """PAML-style sequential PHYLIP: header ``n  length [opts]``, then for each
        taxon a name line followed by sequence lines (blanks inside lines allowed)."""
        if seqtype != "codon":
            raise ValueError("only codon alignments are on the hot path")
        lines = [ln.strip() for ln in Path(path).read_text().splitlines()]
        head = lines[0].split()
        n_species, n_chars = int(head[0]), int(head[1])
        if n_chars % 3:
            raise ValueError(f"Codon sequence length {n_chars} not divisible by 3")
        names, seqs = [], []
        i = 1
        while i < len(lines) and len(names) < n_species:
            if not lines[i]:
                i += 1
                continue
            names.append(lines[i])
            i += 1
            buf = ""
            while i < len(lines) and len(buf) < n_chars:
                buf += re.sub(r"\s", "", lines[i]).upper()
                i += 1
            if len(buf) != n_chars:
                raise ValueError(f"Sequence {names[-1]} has length {len(buf)}, expected {n_chars}")
            seqs.append(buf)
        if len(names) != n_species:
            raise ValueError(f"Expected {n_species} sequences, found {len(names)}")
        return cls(names, np.stack([encode_codon_string(s) for s in seqs]))

    @classmethod
    def from_fasta(cls, path, seqtype: str = "codon") -> "Alignment":
        if seqtype != "codon":
            raise ValueError("only codon alignments are on the hot path")
        names, chunks = [], []
        for ln in Path(path).read_text().splitlines():
            ln = ln.strip()
            if not ln:
                continue
            if ln.startswith(">"):
                names.append(ln[1:].strip())
                chunks.append([])
            elif names:
                chunks[-1].append(re.sub(r"\s", "", ln).upper())
        if not names:
            raise ValueError("No sequences found in FASTA file")
        seqs = ["".join(c) for c in chunks]
        if len({len(s) for s in seqs}) != 1:
            raise ValueError("Sequences have different lengths")
        if len(seqs[0]) % 3:
            raise ValueError(f"Codon sequence length {len(seqs[0])} not divisible by 3")
        return cls(names, np.stack([encode_codon_string(s) for s in seqs]))

    @classmethod
    def from_codes(cls, names, codes) -> "Alignment":
        return cls(list(names), np.asarray(codes))


@dataclass
class Tree:
    """Rooted tree as flat arrays, nodes numbered in Newick reading order (pre-order).

    parent[v] = -1 for the root. ``children[v]`` keeps Newick order.  ``label[v]``
    is the integer after ``#`` on the branch above v (0 when absent).
    """

    parent: np.ndarray
    children: list
    name: list
    branch_length: np.ndarray
    label: np.ndarray
    has_label: np.ndarray = field(default=None)

    @property
    def n_nodes(self) -> int:
        return len(self.parent)

    @property
    def root(self) -> int:
        return int(np.flatnonzero(self.parent < 0)[0])

    def is_leaf(self, v: int) -> bool:
        return not self.children[v]

    @property
    def leaves(self) -> list:
        return [v for v in self.postorder() if not self.children[v]]

    @property
    def n_leaves(self) -> int:
        return sum(1 for c in self.children if not c)

    @property
    def leaf_names(self) -> list:
        return [self.name[v] if self.name[v] else str(v) for v in self.preorder() if not self.children[v]]

    def preorder(self) -> list:
        out, stack = [], [self.root]
        while stack:
            v = stack.pop()
            out.append(v)
            stack.extend(reversed(self.children[v]))
        return out

    def postorder(self) -> list:
        """Children (in Newick order) before their parent; iterative, so deep trees are fine."""
        out, stack = [], [(self.root, 0)]
        while stack:
            v, k = stack.pop()
            if k < len(self.children[v]):
                stack.append((v, k + 1))
                stack.append((self.children[v][k], 0))
            else:
                out.append(v)
        return out

    def branches(self) -> list:
        """(parent, child) pairs in pre-order -- the order PAML prints branch lengths in."""
        return [(int(self.parent[v]), v) for v in self.preorder() if self.parent[v] >= 0]

    def branch_labels(self) -> list:
        return [int(self.label[c]) for _, c in self.branches()]

    def set_branch_lengths_preorder(self, values) -> None:
        for (_, c), t in zip(self.branches(), values):
            self.branch_length[c] = float(t)

    # -- Newick ----------------------------------------------------------
    @classmethod
    def from_newick(cls, text: str) -> "Tree":
        text = re.sub(r"//.*", "", text)
        text = re.sub(r"/\s*\*.*?\*\s*/", "", text)
        # first line-run that holds a tree: skip PAML "n_taxa n_trees" header lines
        kept = []
        for ln in text.split("\n"):
            if not ln.strip() or re.match(r"^\s*\d+\s+\d+\s*$", ln):
                continue
            kept.append(ln)
            if ";" in ln:
                break
        s = "".join(kept).replace("\t", "").replace("\r", "")
        if ";" not in s:
            raise ValueError("Invalid Newick format: missing semicolon")
        s = s[: s.index(";") + 1]

        parent, children, name, blen, label, has_label = [], [], [], [], [], []

        def new_node(par):
            parent.append(par)
            children.append([])
            name.append(None)
            blen.append(0.0)
            label.append(0)
            has_label.append(False)
            v = len(parent) - 1
            if par >= 0:
                children[par].append(v)
            return v

        pos, n = 0, len(s)
        cur = new_node(-1)
        # `cur` is the node whose description we are inside of
        open_stack = []
        tok_end = set(",:();# \n")
        while pos < n:
            ch = s[pos]
            if ch in " \n":
                pos += 1
            elif ch == "(":
                open_stack.append(cur)
                cur = new_node(cur)
                pos += 1
            elif ch == ",":
                cur = new_node(open_stack[-1])
                pos += 1
            elif ch == ")":
                if not open_stack:
                    raise ValueError(f"Unbalanced ')' at position {pos}")
                cur = open_stack.pop()
                pos += 1
            elif ch == "#":
                q = pos + 1
                while q < n and s[q] not in tok_end:
                    q += 1
                try:
                    label[cur] = int(s[pos + 1:q])
                except ValueError:
                    raise ValueError(f"Invalid branch label: #{s[pos + 1:q]}")
                has_label[cur] = True
                pos = q
            elif ch == ":":
                q = pos + 1
                while q < n and s[q] in " \n":
                    q += 1
                r = q
                while r < n and s[r] not in ",(); \n":
                    r += 1
                try:
                    blen[cur] = float(s[q:r])
                except ValueError:
                    raise ValueError(f"Invalid branch length: {s[q:r]}")
                pos = r
            elif ch == ";":
                break
            else:
                q = pos
                while q < n and s[q] not in tok_end:
                    q += 1
                name[cur] = s[pos:q]
                pos = q
        if open_stack:
            raise ValueError("Unbalanced '(' in Newick string")
        return cls(np.array(parent, dtype=np.int64), children, name,
                   np.array(blen, dtype=np.float64), np.array(label, dtype=np.int64),
                   np.array(has_label, dtype=bool))

    def to_newick(self) -> str:
        def rec(v):
            if self.children[v]:
                s = "(" + ",".join(rec(c) for c in self.children[v]) + ")" + (self.name[v] or "")
            else:
                s = self.name[v] or ""
            if self.has_label is not None and self.has_label[v]:
                s += f"#{int(self.label[v])}"
            if self.parent[v] >= 0:
                s += f":{float(self.branch_length[v])!r}"
            return s
        return rec(self.root) + ";"
