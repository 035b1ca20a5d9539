"""Per-category answer accuracies of AGQA with the reference evaluator's interface and result order
(AGQA/src/tasks/agqa_data.py:341-1146: AGQAEvaluator.evaluateOverall / evaluateAllQtypes / evaluateCompSteps /
evaluateNovelComp / evaluateIndirectRef / evaluatePrecision / dump_result).

The reference spells every counter out by hand (about 800 lines); here a metric is a ROW of a table -
(name, predicate over the question's annotation) - and one tally routine produces hits / count for each row in
table order, which is the order of the reference's result lists (its __main__ prints them by position,
agqaHGQA.py:886-1040).  A question's annotation (`datum`) carries the reference's fields: answer, ans_type
('binary' | 'open'), global (list of reasoning types), semantic, structural, nc_seq / nc_sup / nc_dur / nc_objrel,
i_obj / i_act / i_temp, indirect, direct_equiv.

A predicate returns how often a question counts for its row: the reasoning types of `global` are tallied per OCCURRENCE
(the reference loops `for q in qType`, agqa_data.py:520-560, so a type listed twice counts twice), everything else 0 / 1.
`tests/golden/evaluator_2k.json` holds every result list of the reference's own class on a 2 000-question set.

One deliberate difference: a category without questions yields float('nan') here, where the reference divides by
zero and raises (its splits always populate every category; synthetic splits need not).
"""
import json
import math


def _is(field, value):
    return lambda d: 1 if d.get(field) == value else 0


def _both(p, q):
    return lambda d: p(d) if q(d) else 0


def _has(qtype):
    return lambda d: sum(1 for q in (d.get("global") or ()) if q == qtype)


_BIN, _OPEN = _is("ans_type", "binary"), _is("ans_type", "open")


def _tri(p):
    """A category plus its binary / open halves."""
    return [p, _both(p, _BIN), _both(p, _OPEN)]


# evaluateAllQtypes (agqa_data.py:363-700): 31 rows
ALL_QTYPES = (
    [("overall", lambda d: 1), ("binary", _BIN), ("open", _OPEN)]
    + list(zip(("object-relationship", "object-relationship binary", "object-relationship open"), _tri(_has("obj-rel"))))
    + [("relationship-action", _has("rel-act")), ("object-action", _has("obj-act"))]
    + list(zip(("superlative", "superlative binary", "superlative open"), _tri(_has("superlative"))))
    + list(zip(("sequencing", "sequencing binary", "sequencing open"), _tri(_has("sequencing"))))
    + [("exists", _has("exists"))]
    + list(zip(("duration-comparison", "duration-comparison binary", "duration-comparison open"), _tri(_has("duration-comparison"))))
    + [("action-recognition", _has("action-recognition"))]
    + list(zip(("object", "object binary", "object open"), _tri(_is("semantic", "object"))))
    + [("relationship", _is("semantic", "relation"))]
    + list(zip(("action", "action binary", "action open"), _tri(_is("semantic", "action"))))
    + [(s, _is("structural", s)) for s in ("query", "compare", "choose", "logic", "verify")]
)
COMP_STEPS = [("overall", lambda d: 1), ("overall binary", _BIN), ("overall open", _OPEN)]              # :702-733
NOVEL_COMP = (COMP_STEPS                                                                                     # :737-883
              + list(zip(("sequencing", "sequencing binary", "sequencing open"), _tri(_is("nc_seq", 1))))
              + list(zip(("superlative", "superlative binary", "superlative open"), _tri(_is("nc_sup", 1))))
              + list(zip(("duration", "duration binary", "duration open"), _tri(_is("nc_dur", 1))))
              + list(zip(("object relationship", "object relationship binary", "object relationship open"), _tri(_is("nc_objrel", 1)))))
INDIRECT = (list(zip(("object", "object binary", "object open"), _tri(_is("i_obj", 1))))                     # :886-1098
            + list(zip(("action", "action binary", "action open"), _tri(_is("i_act", 1))))
            + list(zip(("localization", "localization binary", "localization open"), _tri(_is("i_temp", 1)))))


def tally(rows, items):
    """items: iterable of (datum, correct: bool) -> [hits / count for every (name, predicate) row], nan when count == 0."""
    preds = [p for _, p in rows]
    hits, cnt = [0] * len(preds), [0] * len(preds)
    for datum, correct in items:
        for i, p in enumerate(preds):
            w = p(datum)
            if w:
                cnt[i] += w
                hits[i] += w if correct else 0
    return [h / c if c else math.nan for h, c in zip(hits, cnt)]


class AGQAEvaluator:
    """dataset: anything with `id2datum` {question id: annotation} and `answerVocab` {answer string: index}
    (agqa_data.py:343-347)."""

    def __init__(self, dataset):
        self.dataset = dataset
        self.answerVocab = dataset.answerVocab
        self.index_to_ans = list(self.answerVocab.keys())

    def _items(self, quesid2ans):
        for qid, ans in quesid2ans.items():
            datum = self.dataset.id2datum[qid]
            yield datum, self.index_to_ans[int(ans)] == datum["answer"]

    def evaluateOverall(self, quesid2ans):
        return tally(ALL_QTYPES[:1], self._items(quesid2ans))[0] if quesid2ans else 0.0

    evaluate = evaluateOverall                       # (oracle_score calls evaluator.evaluate, agqaHGQA.py:856)

    def evaluateAllQtypes(self, quesid2ans):
        return tally(ALL_QTYPES, self._items(quesid2ans))

    def evaluateCompSteps(self, quesid2ans):
        return tally(COMP_STEPS, self._items(quesid2ans))

    def evaluateNovelComp(self, quesid2ans):
        return tally(NOVEL_COMP, self._items(quesid2ans))

    def evaluateIndirectRef(self, quesid2ans):
        """-> (recall list [9], precision questions): an indirect-reference question whose direct equivalent was answered
        correctly goes on to evaluatePrecision with its own prediction attached (agqa_data.py:1066-1086)."""
        recall = tally(INDIRECT, self._items(quesid2ans))
        precision_qs = []
        for qid, ans in quesid2ans.items():
            datum = self.dataset.id2datum[qid]
            eq = datum.get("direct_equiv")
            if eq is None or datum.get("indirect") != 1 or eq not in self.dataset.id2datum or eq not in quesid2ans:
                continue
            if self.index_to_ans[int(quesid2ans[eq])] == self.dataset.id2datum[eq]["answer"]:
                q = dict(datum)
                q["prediction"] = self.index_to_ans[int(ans)]
                precision_qs.append(q)
        return recall, precision_qs

    def evaluatePrecision(self, questions):
        return tally(INDIRECT, ((q, q["prediction"] == q["answer"]) for q in questions))

    def dump_result(self, quesid2ans, path):
        """One record per question with the fields the reference writes (agqa_data.py:1107-1146)."""
        out = []
        for qid, ans in quesid2ans.items():
            d = self.dataset.id2datum[qid]
            rec = {"id": d.get("question_id", qid), "question": d.get("question"), "ans_type": d.get("ans_type"),
                   "question type": d.get("global"), "answer": d["answer"], "prediction": self.index_to_ans[int(ans)],
                   "directEq": d.get("direct_equiv")}
            rec.update({k: d.get(k) for k in ("i_obj", "i_act", "i_temp", "indirect", "nc_seq", "nc_sup", "nc_dur", "nc_objrel",
                                               "semantic", "structural")})
            out.append(rec)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
        return path


def format_report(title, rows, values):
    """The text block the reference's __main__ prints for a result list (agqaHGQA.py:886-1040): one `name: percent` line per row."""
    lines = [title]
    for (name, _), v in zip(rows, values):
        lines.append("%s: %s" % (name, "n/a" if (isinstance(v, float) and math.isnan(v)) else "%0.2f" % (100.0 * v)))
    return "\n".join(lines)
