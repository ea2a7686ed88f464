"""Host side of the hot path's INPUT contract: caption tokenizer, vocabulary and the equal-length minibatcher.

Mirror of `tokenizer.jl` (all) and `lrcn.jl:248-327` (`initeosbos`, `minibatch`, `delete_unbatchable_captions!`) --
SURVEY.md 8(f) row 1.  Pure host code (strings and integers); nothing here touches the GPU.  The output is exactly what
`loss` / `train1` consume: `(sequence, input_ids, lengths)` with the reference's 1-based word ids, plus `batches()` which
cuts it into the `[T][B]` 0-based int32 token blocks of the C ABI (include/lrcn.h).

Deliberate, documented differences from the reference:
  * vocabulary order: the reference's ids follow Julia `Dict` iteration order (`tokenizer.jl:161-163`), which is not
    reproducible; here words get ids in order of first occurrence (specials first: `~~`=eos=1, ` `` `=bos=2, `##`=unk=3);
  * the Flickr val/test split follows Julia's `srand(5); shuffle` (`tokenizer.jl:59-63`); here it is an explicit list of
    line numbers, a NumPy permutation with seed 5, or the image ids of a committed test set (`eval/ids_flickr_bm5`);
  * `delete_unbatchable_captions!` cannot spin (`lrcn.jl:311-318`): the one input on which the reference's scan never ends
    (unsorted lengths: a batch window that straddles the longest length) raises ValueError here;
  * `delete_unbatchable_captions(..., reference_tail=False)` is this repo's saner variant (keep, per length, the largest
    multiple of the batch size).  The DEFAULT reproduces the reference, whose scan always ends by deleting everything from
    its cursor on once fewer than a batch is left behind it (`lrcn.jl:320-323`) -- i.e. the last <= batch_size captions go even
    when they form a complete equal-length batch ([3]*20 at batch 10 keeps 10, not 20).
"""
import json
import re

import numpy as np

EOS, BOS, UNK = 1, 2, 3  # reference ids (lrcn.jl:248-255); the C ABI uses id - 1
EOS_WORD, BOS_WORD, UNK_WORD = "~~", "``", "##"
_STRIP = " .,#')(!/?\t`"              # tokenizer.jl:42, 96, 118
_FLICKR_SPLIT = re.compile(r"[ \t#.\n]")  # tokenizer.jl:37, 91


def _clean(words):
    out = []
    for w in words:
        w = w.strip(_STRIP).lower()
        if w:
            out.append(w)
    return out


def tokenize_flickr_line(line):
    """One line of results_20130124.token: '<id>.jpg#<k>\\t<caption>' -> (image id, words).  (tokenizer.jl:36-48)
    The split on ' ', tab, '#', '.', newline makes fields 1..3 = id, 'jpg', caption number; words start at field 4."""
    fields = [f.lower() for f in _FLICKR_SPLIT.split(line.rstrip("\n"))]
    return int(fields[0]), _clean(fields[3:])


def tokenize_flickr(lines, sort=True):
    """-> [((id, words), length)] stably sorted by length.  (tokenize_flicker_captions / flicker, tokenizer.jl:34-53, 88-109)"""
    caps = []
    for line in lines:
        if not line.strip():
            continue
        i, w = tokenize_flickr_line(line)
        caps.append(((i, w), len(w)))
    if sort:
        caps.sort(key=lambda t: t[1])  # Python's sort is stable, like Julia's default for sort(by=...)
    return caps


def split_flickr(lines, val_size=1000, test_size=1000, seed=5, test_ids=None, val_ids=None):
    """Train / val / test caption lists (tokenize_flicker_captions2, tokenizer.jl:56-87).  Images are groups of 5
    consecutive lines.  Selection: `test_ids` / `val_ids` (image ids, e.g. the committed eval/ids_flickr_bm5) when
    given, else a seeded permutation of the image groups."""
    lines = [ln for ln in lines if ln.strip()]
    starts = list(range(0, len(lines) - len(lines) % 5, 5))
    if test_ids is not None or val_ids is not None:
        tset, vset = set(test_ids or ()), set(val_ids or ())
        val_g = [s for s in starts if tokenize_flickr_line(lines[s])[0] in vset]
        test_g = [s for s in starts if tokenize_flickr_line(lines[s])[0] in tset]
    else:
        perm = np.random.default_rng(seed).permutation(len(starts))
        val_g = sorted(starts[i] for i in perm[:val_size])
        test_g = sorted(starts[i] for i in perm[val_size:val_size + test_size])
    held = set()
    for s in val_g + test_g:
        held.update(range(s, s + 5))
    pick = lambda groups: [lines[s + k] for s in sorted(groups) for k in range(5)]  # noqa: E731
    train = [ln for i, ln in enumerate(lines) if i not in held]
    return tokenize_flickr(train), tokenize_flickr(pick(val_g)), tokenize_flickr(pick(test_g))


def tokenize_coco(json_text):
    """captions_{train,val}2014.json -> [((image_id, words), length)] sorted by length.  (tokenizer.jl:111-130)"""
    caps = []
    for obj in json.loads(json_text)["annotations"]:
        w = _clean(obj["caption"].split(" "))
        caps.append(((obj["image_id"], w), len(w)))
    caps.sort(key=lambda t: t[1])
    return caps


def build_vocab(caption_lists, threshold=5):
    """Words seen >= `threshold` times over all lists (the reference counts Flickr val/test captions too,
    tokenizer.jl:13-15, 147-152), specials first.  -> dict word -> 1-based id.  (get_vocab + filtervocab, :132-166)"""
    counts, order = {}, []
    for caps in caption_lists:
        for (_, words), _n in caps:
            for w in words:
                if w not in counts:
                    counts[w] = 0
                    order.append(w)
                counts[w] += 1
    vocab = {EOS_WORD: EOS, BOS_WORD: BOS, UNK_WORD: UNK}
    for w in order:
        if counts[w] >= threshold and w not in vocab:
            vocab[w] = len(vocab) + 1
    return vocab


def index_to_word(vocab):
    out = [None] * len(vocab)
    for w, i in vocab.items():
        out[i - 1] = w
    return out


def reference_delete_ranges(lengths, batch_size):
    """The 0-based indices `delete_unbatchable_captions!` removes (lrcn.jl:299-327), by running ITS loop on `lengths` -- the
    cursor arithmetic below is the reference's, 1-based, statement for statement (integer work: the bar is identical output):
      :301 limit = n - B + 1          :304 cursor = 1, current_length = lengths[1]
      :306 a full window of the current length -> cursor += B
      :308-318 otherwise the cursor jumps to the first caption of the next length PRESENT (findfirst; lengths that do not occur
               are skipped, :311-317) and [old cursor, new cursor) is deleted
      :320-323 as soon as cursor >= limit everything from the cursor to the end is deleted and the scan stops.
    The loop can only be left through :320-323, so the tail [cursor, n] -- between 1 and B captions -- is ALWAYS deleted.
    n < B: the reference's loop body never runs and nothing is deleted (minibatch would then index out of bounds, :283);
    here that is an empty result (every caption is unbatchable)."""
    n, B = len(lengths), int(batch_size)
    if B <= 0:
        raise ValueError("batch_size must be positive")
    if n == 0:
        return []
    if n < B:
        return list(range(n))
    limit = n - B + 1
    max_length = max(lengths)
    first = {}
    for i, v in enumerate(lengths):          # findfirst(lengths, v), 1-based
        first.setdefault(v, i + 1)
    current_length, cur = lengths[0], 1
    ranges = []
    while cur < limit:
        if lengths[cur + B - 2] == current_length:          # lengths[cur+B-1], 1-based
            cur += B
        else:
            old, cur = cur, 0
            while cur == 0:
                current_length += 1
                if current_length > max_length:
                    break
                cur = first.get(current_length, 0)
            if cur == 0:
                # lrcn.jl:311-318 leaves cursor = 0 here and the outer loop never ends; unreachable for sorted lengths
                raise ValueError("delete_unbatchable_captions: lengths are not sorted (the reference's scan does not terminate on this input)")
            ranges.extend(range(old, cur))
        if cur >= limit:
            ranges.extend(range(cur, n + 1))
            break
    return [i - 1 for i in ranges]


def delete_unbatchable_captions(caps, batch_size, reference_tail=True):
    """`delete_unbatchable_captions!` (lrcn.jl:299-327): drop the captions that cannot sit in an equal-length batch.  `caps` must
    be sorted by length.  Returns a new list.
    reference_tail=True (default): exactly the reference's result, see reference_delete_ranges -- per length the largest
    multiple of `batch_size` survives, EXCEPT that the scan's last window is always deleted (the final <= batch_size captions).
    reference_tail=False: per length the largest multiple of `batch_size`, nothing else dropped."""
    if reference_tail:
        gone = set(reference_delete_ranges([c[1] for c in caps], batch_size))
        return [c for i, c in enumerate(caps) if i not in gone]
    out, i, n = [], 0, len(caps)
    while i < n:
        j = i
        while j < n and caps[j][1] == caps[i][1]:
            j += 1
        keep = (j - i) // batch_size * batch_size
        out.extend(caps[i:i + keep])
        i = j
    return out


def minibatch(caps, word_to_index, batch_size, reference_tail=True):
    """-> (sequence, input_ids, lengths) exactly as lrcn.jl:257-297: `sequence[k]` is the vector of the k-th word of a
    batch (batches concatenated along k), `input_ids[b]` the image ids of batch b, `lengths` the per-caption lengths of
    the kept captions.  Splits with <= 30000 captions are forced to batch 10 (lrcn.jl:260-270).  Returns the batch size
    actually used as a fourth value.
    Sizing (lrcn.jl:276-281): the reference allocates `nbatch = div(sum(lengths), batch_size)` word vectors and one id vector
    per `1:batch_size:length(lengths)`; after the deletion every length group is a whole number of batches, so
    sum(lengths) / batch_size = sum over batches of T -- the number of rows appended below (asserted)."""
    if len(caps) <= 30000:
        batch_size = 10
    caps = delete_unbatchable_captions(caps, batch_size, reference_tail=reference_tail)
    lengths = [c[1] for c in caps]
    sequence, input_ids = [], []
    for i in range(0, len(caps), batch_size):
        group = caps[i:i + batch_size]
        T = group[0][1]
        input_ids.append([g[0][0] for g in group])
        for k in range(T):
            sequence.append([word_to_index.get(g[0][1][k], UNK) for g in group])
    assert len(sequence) == sum(lengths) // batch_size and len(input_ids) == len(range(0, len(lengths), batch_size))
    return sequence, input_ids, lengths, batch_size


def batches(sequence, input_ids, lengths, batch_size, max_len=28):
    """Iterate the minibatches the way train1 / average_loss index them (lrcn.jl:351-378, 436-452): yields
    (image ids, tokens) with tokens an int32 [T][B] array of 0-based ids (the C ABI's convention); captions longer than
    `max_len` words are skipped like the reference does (lrcn.jl:353-355)."""
    k = 0
    for b, ids in enumerate(input_ids):
        T = lengths[b * batch_size]
        block = sequence[k:k + T]
        k += T
        if T > max_len:
            continue
        yield ids, (np.asarray(block, dtype=np.int32).reshape(T, batch_size) - 1)


def caption_text(token_ids, idx2word):
    """generate()'s output line (lrcn.jl:634-640): words after bos up to the first eos, space separated, then '.'.
    `token_ids` are 0-based ABI ids as returned by lrcn_beam_search (bos first)."""
    words = []
    for t in list(token_ids)[1:]:
        if t + 1 == EOS:
            break
        words.append(idx2word[t])
    return " ".join(words + ["."]) if words else "."
