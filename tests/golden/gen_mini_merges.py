"""Train a tiny byte-pair merge list (format of the reference's bpe_simple_vocab_16e6.txt: a header line,
then one "left right" pair per line) for tokenizer tests that must run without the reference's file.

    python tests/golden/gen_mini_merges.py      # writes tests/golden/mini_merges.txt (every merge the corpus supports, 264)

The corpus is the text below; symbols are the byte-level alphabet of src/tokenizer.rs:6-28 and the last
symbol of a word carries "</w>", exactly as the real merges file was built.
"""
import collections
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parents[1]))
from oracle.tokenizer_oracle import bytes_to_unicode  # noqa: E402

CORPUS = """
a photo of an astronaut riding a horse on mars. a painting of a cat sitting on a sofa, highly detailed, trending on
artstation. an oil painting of the sea at night, the moon and the stars above the water. a portrait of a woman with
red hair, soft light, studio photo. the quick brown fox jumps over the lazy dog. it's a dog's life, isn't it? they're
here and we've seen what you'd do. a cyberpunk city at night, neon lights, rain, reflections on the street, 4k, 8k,
35mm, f/1.8. a watercolor of mountains and a lake, autumn colors. hello world! hello again, world. 1234567890
cafe naive resume uber strasse. the astronaut's horse is riding the astronaut. painting, paintings, painted, painter.
""" * 3

N_MERGES = 400   # upper bound; the corpus runs out first


def main():
    enc = dict(bytes_to_unicode())
    words = collections.Counter()
    for w in CORPUS.lower().split():
        sym = [enc[b] for b in w.encode("utf-8")]
        sym[-1] += "</w>"
        words[tuple(sym)] += 1
    merges = []
    for _ in range(N_MERGES):
        pairs = collections.Counter()
        for w, c in words.items():
            for a, b in zip(w, w[1:]):
                pairs[(a, b)] += c
        if not pairs:
            break
        best = max(sorted(pairs), key=lambda p: pairs[p])   # deterministic tie-break
        merges.append(best)
        new_words = collections.Counter()
        for w, c in words.items():
            out, i = [], 0
            while i < len(w):
                if i + 1 < len(w) and (w[i], w[i + 1]) == best:
                    out.append(w[i] + w[i + 1])
                    i += 2
                else:
                    out.append(w[i])
                    i += 1
            new_words[tuple(out)] += c
        words = new_words
    with open(HERE / "mini_merges.txt", "w", encoding="utf-8") as f:
        f.write("#version: mini\n")
        for a, b in merges:
            f.write(f"{a} {b}\n")
    print(f"wrote {len(merges)} merges")


if __name__ == "__main__":
    main()
