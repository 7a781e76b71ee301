"""The C++ byte-level BPE reader (csrc/q3_tokenizer.cpp, through the C ABI) against the reference's own tokenizer library.

The reference's Tokenizer (src/utils/tokenizer.rs:1-37) is a wrapper over the `tokenizers` crate 0.22 (Cargo.toml:20):
`encode(text, add_special_tokens = false).get_ids()`. The Python package `tokenizers` 0.22.x in this image is a binding of
that same crate, so here parity IS pinned: ids must be identical. The real model's tokenizer.json is not available offline;
the tests train byte-level BPE tokenizers with the Qwen2 pipeline (NFC, the Qwen2 split regex, ByteLevel, special added
tokens) on a small multilingual corpus and compare on text that exercises every branch of the regex. Host only.
"""
import json

import numpy as np
import pytest

tokenizers = pytest.importorskip("tokenizers")

QWEN2_PAT = r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+"

CORPUS = [
    "Hello world, it's a test of the tokenizer's merges 12345 and 67890.",
    "你好，世界！今天天气怎么样？我们一起去公园散步吧。语音合成系统需要分词器。",
    "The quick brown fox\n\njumps over  the lazy dog\r\n\r\nand keeps running...",
    "émigré naïve café résumé 2024年10月4日 ３２１ ²³ Ⅷ",
    "  leading spaces and\ttabs\t\t trailing   ",
    "I'LL we'Ve DON'T you're he'd she's I'm THEY'RE",
    "def f(x): return x**2 + 3*x - 1  # comment\n\tif x != 0 && y >= 1 { z <<= 2; }",
    "Привет, мир! こんにちは世界。안녕하세요 세계. مرحبا بالعالم",
    "emoji 😀😀 🎉 and symbols ©®™ §¶ • … — “quotes” ‘single’",
    "a b c　d   ef",
]


def _train(vocab_size, merges_as_strings=False, tmp_path=None, name="tokenizer.json"):
    from tokenizers import Regex, Tokenizer, decoders, models, normalizers, pre_tokenizers, trainers
    tok = Tokenizer(models.BPE())
    tok.normalizer = normalizers.NFC()
    tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Split(Regex(QWEN2_PAT), behavior="isolated", invert=False),
                                                 pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
    tok.decoder = decoders.ByteLevel()
    tr = trainers.BpeTrainer(vocab_size=vocab_size, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(),
                             special_tokens=["<|endoftext|>", "<|im_start|>", "<|im_end|>", "<|im_start|>assistant"], show_progress=False)
    tok.train_from_iterator(CORPUS * 8, tr)
    path = tmp_path / name
    tok.save(str(path))
    if merges_as_strings:  # the older on-disk form: "a b" strings instead of ["a", "b"] pairs
        j = json.loads(path.read_text(encoding="utf-8"))
        j["model"]["merges"] = [m if isinstance(m, str) else " ".join(m) for m in j["model"]["merges"]]
        path.write_text(json.dumps(j, ensure_ascii=True), encoding="utf-8")  # \\uXXXX escapes and surrogate pairs for the parser
        tok = Tokenizer.from_file(str(path))
    return tok, path


TEXTS = CORPUS + [
    "", " ", "\n", "  \n  \n", "x", "'", "''s", "'S 'T 'Re 'vE 'M 'lL 'D 'x", "it'sſ 'ſ",
    "word  two   three    four\n", "tail space ", "tail spaces   ", "   ", " \t\n\r\n x",
    "<|im_start|>user\n你好<|im_end|>\n<|im_start|>assistant\n", "<|im_start|>assistantx <|endoftext|><|endoftext|>", "<|im_start", "a<|im_end|>b",
    "!!!???...\n\nnext", " !x", "  !x", "1a2b3c 42 3.14159 1,000,000", "x²+y³=z⁴", "\r\r\n\n\r", "a\r\nb", "tab\there", "line1\n line2\n  line3",
    "MiXeD CaSe's TeXt'LL", "数字123和letters混合text", "𝒳𝒴𝒵 𐍈 \U0001F600", "don't’t", "end.\n", "...\r\n", "   　x",
    "The reference encodes text like this: 今天天气真好，我们去公园玩吧！",
]


@pytest.mark.parametrize("vocab_size,strings", [(420, False), (900, True)])
def test_ids_equal_the_tokenizers_crate(tmp_path, vocab_size, strings):
    from q3tts import native
    hf, path = _train(vocab_size, strings, tmp_path)
    tk = native.NativeTokenizer(path)
    try:
        assert tk.vocab_size == hf.get_vocab_size(with_added_tokens=True)
        for text in TEXTS:
            want = hf.encode(text, add_special_tokens=False).ids
            got = tk.encode(text).tolist()
            assert got == want, (text, got, want)
            assert tk.decode(got) == hf.decode(want, skip_special_tokens=False), text
        # seeded random strings over an alphabet that mixes every class the regex distinguishes
        rng = np.random.default_rng(vocab_size)
        alphabet = list("abcXYZ'stredvml 019\n\r\t.,!?-_()你好世界éñ😀 　") + ["<|im_end|>", "  ", "'ll", "'RE"]
        for _ in range(400):
            text = "".join(rng.choice(alphabet, size=int(rng.integers(1, 40))))
            assert tk.encode(text).tolist() == hf.encode(text, add_special_tokens=False).ids, repr(text)
        long_text = " ".join(CORPUS) * 20
        assert tk.encode(long_text).tolist() == hf.encode(long_text, add_special_tokens=False).ids
    finally:
        tk.close()


def test_nfc_normaliser_matches_unicodedata_and_the_crate(tmp_path):
    """The file's normaliser is NFC (UAX #15). ByteLevel BPE is lossless, so decode(encode(x)) is the normalised text: it must
    equal unicodedata.normalize("NFC", x) (the tables are generated from the same data), and the ids must equal the crate's
    on combining sequences, reordering of marks, Hangul jamo, singletons and composition exclusions."""
    import unicodedata as ud
    from q3tts import native
    hf, path = _train(420, False, tmp_path)
    tk = native.NativeTokenizer(path)
    try:
        samples = ["e\u0301", "A\u030a ngstro\u0308m", "a\u0323\u0301 vs a\u0301\u0323", "\u1100\u1161\u11a8 \u1112\u1161\u11ab\u1100\u1173\u11af", "\u212b \u2126 \u00c5",
                   "\u0958 \u0915\u093c", "\u1e9b\u0323", "q\u0307\u0323", "\u0301leading mark", "\u0041\u0300\u0301\u0302\u0303", "cafe\u0301 nai\u0308ve re\u0301sume\u0301",
                   "\ud55c\uae00 plain \u4f60\u597d", "\u0f73\u0f75\u0f81", "\u0b47\u0b56 \u0b47\u0b3e \u0b47\u0b57", "D\u0307\u0323 d\u0323\u0307"]
        for x in samples:
            want = ud.normalize("NFC", x)
            ids = tk.encode(x).tolist()
            assert tk.decode(ids) == want, (x.encode("unicode_escape"), tk.decode(ids).encode("unicode_escape"), want.encode("unicode_escape"))
            assert ids == hf.encode(x, add_special_tokens=False).ids, x.encode("unicode_escape")
        # seeded random strings over starters, marks of several combining classes, jamo and precomposed letters
        pool = [chr(c) for c in [0x61, 0x65, 0x6F, 0x41, 0xE9, 0xC5, 0x1E9B, 0x300, 0x301, 0x302, 0x308, 0x30A, 0x323, 0x327, 0x328, 0x334, 0x345, 0x1100, 0x1161, 0x11A8,
                                 0xAC00, 0xAC01, 0x915, 0x93C, 0x20, 0x4F60, 0x212B, 0x3099, 0x304B, 0x5B0, 0x5B4]]
        rng = np.random.default_rng(7)
        for _ in range(500):
            x = "".join(rng.choice(pool, size=int(rng.integers(1, 12))))
            assert tk.decode(tk.encode(x).tolist()) == ud.normalize("NFC", x), x.encode("unicode_escape")
    finally:
        tk.close()


def test_tokenizer_refuses_what_it_does_not_implement(tmp_path):
    from q3tts import _abi, native
    hf, path = _train(420, False, tmp_path)
    tk = native.NativeTokenizer(path)
    with pytest.raises(_abi.Q3Error, match="UTF-8"):
        tk.encode(b"\xff\xfe")
    tk.close()
    base = json.loads(path.read_text(encoding="utf-8"))

    def variant(mut, name):
        j = json.loads(json.dumps(base))
        mut(j)
        p = tmp_path / name
        p.write_text(json.dumps(j), encoding="utf-8")
        return p
    cases = [
        (lambda j: j["pre_tokenizer"]["pretokenizers"][0]["pattern"].update(Regex=r"\s+"), "Qwen2 pattern"),
        (lambda j: j.update(normalizer={"type": "NFKC"}), "normalizer"),
        (lambda j: j["model"].update(byte_fallback=True), "byte_fallback"),
        (lambda j: j["model"].update(type="WordPiece"), "BPE"),
        (lambda j: j["added_tokens"][0].update(lstrip=True), "lstrip"),
        (lambda j: j["model"]["merges"].append(["zzzz", "qqqq"]), "outside the vocabulary"),
    ]
    for i, (mut, msg) in enumerate(cases):
        with pytest.raises(_abi.Q3Error, match=msg):
            native.NativeTokenizer(variant(mut, f"bad{i}.json"))
    (tmp_path / "trunc.json").write_text(path.read_text(encoding="utf-8")[:2000], encoding="utf-8")
    with pytest.raises(_abi.Q3Error, match="JSON"):
        native.NativeTokenizer(tmp_path / "trunc.json")
    with pytest.raises(_abi.Q3Error, match="Failed to load tokenizer"):
        native.NativeTokenizer(tmp_path / "missing.json")


def test_api_engine_text_path_uses_the_native_tokenizer(tmp_path):
    """TtsEngine::new loads model_dir/tokenizer/tokenizer.json (src/utils/tokenizer.rs:11-13); the mirror encodes text with
    the C++ reader and gets the crate's ids."""
    from q3tts import api
    (tmp_path / "tokenizer").mkdir()
    hf, path = _train(420, False, tmp_path / "tokenizer")
    tok = api.load_tokenizer(str(tmp_path))
    te = api.TtsEngine.__new__(api.TtsEngine)
    te.tokenizer = tok
    text = "你好，世界！it's 2024."
    assert te._encode(text).tolist() == hf.encode(text, add_special_tokens=False).ids
    assert api.load_tokenizer(str(tmp_path / "nowhere")) is None
