"""Word error rate, as the reference's summarize.py computes it with `jiwer.wer(references, hypotheses)`
(W/summarize.py:159-181): one word-level Levenshtein alignment per sentence pair, errors and reference
words summed over the corpus, WER = (S + D + I) / N.  jiwer is not in this image, so the (small) metric
is implemented here."""
from typing import Dict, List, Sequence, Union


def _words(s: str) -> List[str]:
    return s.split()          # jiwer's default transform: strip, collapse blanks, split on spaces


def edit_counts(reference: Sequence[str], hypothesis: Sequence[str]) -> Dict[str, int]:
    """Minimum-edit alignment of two word lists -> {'hits','substitutions','deletions','insertions'}.
    Among alignments of equal cost, ties are broken hit/substitution > deletion > insertion."""
    n, m = len(reference), len(hypothesis)
    # cost[i][j]: edits turning reference[:i] into hypothesis[:j]
    cost = [[0] * (m + 1) for _ in range(n + 1)]
    for i in range(1, n + 1):
        cost[i][0] = i
    for j in range(1, m + 1):
        cost[0][j] = j
    for i in range(1, n + 1):
        ri, row, up = reference[i - 1], cost[i], cost[i - 1]
        for j in range(1, m + 1):
            diag = up[j - 1] + (ri != hypothesis[j - 1])
            row[j] = min(diag, up[j] + 1, row[j - 1] + 1)
    counts = dict(hits=0, substitutions=0, deletions=0, insertions=0)
    i, j = n, m
    while i > 0 or j > 0:
        if i > 0 and j > 0 and cost[i][j] == cost[i - 1][j - 1] + (reference[i - 1] != hypothesis[j - 1]):
            counts["hits" if reference[i - 1] == hypothesis[j - 1] else "substitutions"] += 1
            i, j = i - 1, j - 1
        elif i > 0 and cost[i][j] == cost[i - 1][j] + 1:
            counts["deletions"] += 1
            i -= 1
        else:
            counts["insertions"] += 1
            j -= 1
    return counts


def wer(references: Union[str, Sequence[str]], hypotheses: Union[str, Sequence[str]]) -> float:
    """Corpus-level word error rate of `hypotheses` against `references` (same length lists or two strings)."""
    if isinstance(references, str):
        references = [references]
    if isinstance(hypotheses, str):
        hypotheses = [hypotheses]
    if len(references) != len(hypotheses):
        raise ValueError(f"{len(references)} references but {len(hypotheses)} hypotheses")
    errors = total = 0
    for ref, hyp in zip(references, hypotheses):
        r, h = _words(ref), _words(hyp)
        if not r:
            raise ValueError("one or more references are empty strings")
        c = edit_counts(r, h)
        errors += c["substitutions"] + c["deletions"] + c["insertions"]
        total += len(r)
    return errors / total
