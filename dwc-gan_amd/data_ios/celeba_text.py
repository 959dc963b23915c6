"""Text commands for a (source labels, target labels) pair of the 8 CelebA attributes — the role of the reference's
``data_ios/celeba_text.labels2text`` (reference celeba_text.py:320-325), re-implemented as a small table-driven grammar.

It is NOT a transcription: it does not consume Python's ``random`` stream like the reference's generator and produces its
own sentences (two styles: one clause per attribute, or a description of the target face).  What it shares with the
reference is the contract the text encoder depends on: every emitted token is in the CelebA vocabulary (vocab.py), the
sentence states the target value of every attribute that changes, mentions of unchanged attributes are optional, and the
length stays far below the 80-token pad width (reference celeba_data.py:98).

Attribute order (reference celeba_text.py:6-15): black, blond, brown hair; male; smile; young; eyeglasses; (no) beard.
"""
import random

HAIR = ("black", "blond", "brown")
VERB = ("make", "change", "translate", "modify")
TO = ("to", "into")
MAN, WOMAN = ("boy", "male", "man", "gentleman", "sir"), ("female", "woman", "lady", "miss", "girl")
SMILE, NOSMILE = ("smile", "smiling", "happy", "delighted", "laugh"), ("unsmiling", "unhappy", "serious", "smileless", "solemn")
YOUNG, OLD = ("young", "younger"), ("old", "older", "big age")
GLASSES, BEARD = ("glasses", "eyeglasses", "sunglasses"), ("beard", "moustache", "whiskers", "beards")
PUT_ON, TAKE_OFF = ("wear", "add", "put on", "with"), ("remove", "take off", "without", "no")
KEEP = ("do nothing on the {}", "do not change the {}", "keep {} unchanged", "keep the {} unchanged")


def _pick(rng, seq):
    return seq[rng.randrange(len(seq))]


def _hair(rng, bits):
    names = [HAIR[i] for i in range(3) if bits[i]]
    if not names:
        return "unknown"
    rng.shuffle(names)
    return names[0] if len(names) == 1 else " and ".join([" , ".join(names[:-1]), names[-1]])


def _clauses(rng, src, trg):
    """One clause per attribute that changes; unchanged attributes are mentioned with probability 1/4."""
    out = []
    poss = _pick(rng, ("his" if trg[3] else "her", "the"))

    def unchanged(what):
        if rng.random() < 0.25:
            out.append(_pick(rng, KEEP).format(what))
    if list(src[:3]) != list(trg[:3]):
        out.append(_pick(rng, ("{v} {p} hair {c} {t} {h}", "{v} hair {c} {t} {h}", "{h} hair", "{h} hair {c}")).format(
            v=_pick(rng, VERB), p=poss, c=_pick(rng, ("color", "colour")), t=_pick(rng, TO), h=_hair(rng, trg)))
    else:
        unchanged("hair color")
    if src[3] != trg[3]:
        out.append("{} {} gender {} {}".format(_pick(rng, VERB), poss, _pick(rng, TO), _pick(rng, MAN if trg[3] else WOMAN)))
    else:
        unchanged("gender")
    if src[4] != trg[4]:
        out.append("{} the face {} be {}".format(_pick(rng, VERB), _pick(rng, TO), _pick(rng, SMILE if trg[4] else NOSMILE)))
    else:
        unchanged("smile")
    if src[5] != trg[5]:
        out.append(_pick(rng, ("{} {} face {}".format(_pick(rng, VERB), poss, _pick(rng, YOUNG if trg[5] else OLD)),
                               "{} age".format(_pick(rng, ("decrease", "reduce") if trg[5] else ("increase", "add"))))))
    else:
        unchanged("age")
    if src[6] != trg[6]:
        out.append("{} {}".format(_pick(rng, PUT_ON if trg[6] else TAKE_OFF), _pick(rng, GLASSES)))
    else:
        unchanged("glasses")
    if src[7] != trg[7]:        # attribute 7 is No_Beard: 1 = clean-shaven
        out.append("{} the {}".format(_pick(rng, TAKE_OFF if trg[7] else PUT_ON), _pick(rng, BEARD)))
    else:
        unchanged("beard")
    rng.shuffle(out)
    return out


def _description(rng, trg):
    """The target face in one noun phrase: 'it be a young smiling lady with black hair , with glasses and without beard'."""
    parts = [_pick(rng, ("it be", "this be", "make it")), _pick(rng, ("a", "an")), _pick(rng, YOUNG if trg[5] else OLD),
             _pick(rng, SMILE if trg[4] else NOSMILE), _pick(rng, MAN if trg[3] else WOMAN), "with", _hair(rng, trg), "hair", ",",
             _pick(rng, ("with", "wear") if trg[6] else ("without", "no")), _pick(rng, GLASSES), "and",
             _pick(rng, ("without", "no") if trg[7] else ("with", "wear")), _pick(rng, BEARD)]
    return " ".join(parts)


def labels2text(src_lab, trg_lab, rng=None):
    """Sentence commanding the edit src -> trg.  ``rng``: a ``random.Random`` (default: the module-level generator, which
    a DataLoader worker seeds per worker)."""
    rng = rng or random
    src, trg = [int(v) for v in src_lab], [int(v) for v in trg_lab]
    if src == trg and rng.random() < 0.5:
        return _pick(rng, ("", "do nothing", "do not change anything", "keep everything unchanged"))
    if rng.random() < 0.3:
        text = _description(rng, trg)
    else:
        text = " . ".join(_clauses(rng, src, trg))
    return text + _pick(rng, ("", " .", " !", " ?")) if text else text
