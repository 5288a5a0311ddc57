"""Bounded best-N container and beam record with the ordering semantics of the reference's
utils/top_n.py (which follows im2txt): a min-heap keyed by Beam.score only, so ties are broken
by heap mechanics exactly as `heapq` does -- beam search results depend on that."""
import heapq


class Beam(object):
    """A (partial) caption: token ids, decoder state handle, log-probability, ranking score."""
    __slots__ = ("sentence", "state", "logprob", "score")

    def __init__(self, sentence, state, logprob, score):
        self.sentence, self.state, self.logprob, self.score = sentence, state, logprob, score

    def __lt__(self, other):
        return self.score < other.score

    def __eq__(self, other):
        return self.score == other.score


class TopN(object):
    """Keeps the n largest pushed items (utils/top_n.py:4-43)."""

    def __init__(self, n):
        self._n = n
        self._heap = []

    def size(self):
        return len(self._heap)

    def push(self, item):
        if len(self._heap) < self._n:
            heapq.heappush(self._heap, item)
        else:
            heapq.heappushpop(self._heap, item)

    def extract(self, sort=False):
        """Destructive: returns the kept items (descending when sort=True)."""
        items, self._heap = self._heap, None
        if sort:
            items.sort(reverse=True)
        return items

    def reset(self):
        self._heap = []
