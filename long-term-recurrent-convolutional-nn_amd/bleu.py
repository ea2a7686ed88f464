"""Caption evaluation: the reference's BLEU scoring and reference-file writer.

`multi_bleu` restates `eval/multi-bleu.perl` as the reference ships it: Moses multi-bleu with the **brevity penalty
commented out** (`multi-bleu.perl:137-139`, BP printed as 1.000) and **cumulative** BLEU-1..4 printed
(`:146-168`): clipped n-gram counts against the per-n-gram maximum over the references (`:66-83, 100-110`), closest
reference length with ties to the shorter (`:49-64`), whitespace tokenisation, no lower-casing unless asked (`-lc`).
`write_coco_refs` / `write_flickr_refs` restate `eval/eval.jl:12-36, 44-75`.  Host-only code.
"""
import math
from collections import Counter


def _ngrams(words, n):
    return Counter(tuple(words[i:i + n]) for i in range(len(words) - n + 1))


def multi_bleu(hypotheses, references, lowercase=False):
    """hypotheses: list of strings; references: list (one per reference file) of lists of strings, line-aligned.
    -> dict(bleu=[b1, b2, b3, b4] in percent (cumulative), bp=1.0, ratio, hyp_len, ref_len)."""
    correct, total = [0] * 5, [0] * 5
    len_hyp = len_ref = 0
    for s, hyp in enumerate(hypotheses):
        if lowercase:
            hyp = hyp.lower()
        words = hyp.split()
        closest_diff, closest_len = 9999, 9999
        ref_max = Counter()
        for ref_file in references:
            ref = ref_file[s].lower() if lowercase else ref_file[s]
            rw = ref.split()
            diff = abs(len(words) - len(rw))
            if diff < closest_diff:
                closest_diff, closest_len = diff, len(rw)
            elif diff == closest_diff and len(rw) < closest_len:
                closest_len = len(rw)
            for n in range(1, 5):
                for g, c in _ngrams(rw, n).items():
                    if ref_max[g] < c:
                        ref_max[g] = c
        len_hyp += len(words)
        len_ref += closest_len
        for n in range(1, 5):
            for g, c in _ngrams(words, n).items():
                total[n] += c
                correct[n] += min(c, ref_max.get(g, 0))
    if len_ref == 0:
        return {"bleu": [0.0] * 4, "bp": 0.0, "ratio": 0.0, "hyp_len": 0, "ref_len": 0}
    prec = [(correct[n] / total[n]) if total[n] else 0.0 for n in range(1, 5)]
    logs = [math.log(p) if p > 0 else -9999999999.0 for p in prec]
    bleu = [100.0 * math.exp(sum(logs[:n]) / n) for n in range(1, 5)]  # brevity penalty disabled, as in the reference
    return {"bleu": bleu, "bp": 1.0, "ratio": len_hyp / len_ref, "hyp_len": len_hyp, "ref_len": len_ref}


def format_bleu(r):
    """The script's output line (`multi-bleu.perl:170-178`)."""
    return "BLEU = %.1f/%.1f/%.1f/%.1f (BP=%.3f, ratio=%.3f, hyp_len=%d, ref_len=%d)" % (
        r["bleu"][0], r["bleu"][1], r["bleu"][2], r["bleu"][3], r["bp"], r["ratio"], r["hyp_len"], r["ref_len"])


def read_refs(stem, max_refs=64):
    """Reference files `stem0`, `stem1`, ... (`multi-bleu.perl:21-28`)."""
    import os
    refs = []
    for k in range(max_refs):
        path = "%s%d" % (stem, k)
        if not os.path.exists(path):
            break
        with open(path) as f:
            refs.append([ln.rstrip("\n") for ln in f])
    if not refs:
        raise FileNotFoundError("could not find reference file %s0" % stem)
    return refs


def coco_reference_lines(annotations, candidate_ids, nrefs=5):
    """eval.jl:12-36: per image the first `nrefs` captions, stripped, trailing '.' removed, ' .' appended, lower-cased;
    -> list of `nrefs` lists aligned with candidate_ids."""
    caps = {}
    for item in annotations:
        arr = caps.setdefault(item["image_id"], [])
        if len(arr) == nrefs:
            continue
        cap = item["caption"].strip().strip(".") + " ."
        arr.append(cap.lower())
    return [[caps[i][k].strip() for i in candidate_ids] for k in range(nrefs)]


def flickr_reference_lines(token_lines, candidate_ids, nrefs=5):
    """eval.jl:44-75: '<id>.jpg#k\\t<caption>' -> lower-cased stripped caption; first `nrefs` per image."""
    caps = {}
    for line in token_lines:
        if not line.strip():
            continue
        info = line.split("#")
        i = int(info[0].split(".")[0])
        cap = info[1].split("\t")[1].strip().lower()
        caps.setdefault(i, []).append(cap)
    missing = [i for i in candidate_ids if i not in caps]
    if missing:
        raise KeyError("id is missing in reference: %d" % missing[0])
    return [[caps[i][k].strip() for i in candidate_ids] for k in range(nrefs)]


def write_refs(stem, ref_lines):
    for k, lines in enumerate(ref_lines):
        with open("%s%d" % (stem, k), "w") as f:
            f.write("\n".join(lines) + "\n")
