// Which slices of the gradient arena go to the all-reduce, and when.  Pure bookkeeping (no device calls), so that it can be
// unit-tested on the CPU (tests/bucket_planner_main.cpp, tests/test_dp_cpu.py).
//
// The arena holds the gradient tensors back to back in the order back-propagation is expected to complete them (last
// layer first, biases before weights).  A tensor is complete once every layer contributing to it (its owner and the layers
// sharing the matrix through weightSource) has run its weight-gradient product.  Completed tensors extend a frontier from
// the start of the arena -- a tensor completing out of order is remembered and joins when the frontier reaches it -- and
// whenever the frontier is at least `minElems` ahead of what has been sent, that stretch is one bucket.  finish() sends
// whatever is left, so every step reduces the whole arena exactly once, in disjoint slices, whatever the arrival order.
#pragma once
#include <cstddef>
#include <vector>

namespace rt {

class BucketPlanner {
public:
    // ends[i]: one past the last (padded) element of tensor i; tensor i starts where i-1 ends (tensor 0 at 0).
    // contributors[i]: gradient products per step that add into tensor i (0: it does not learn and never arrives).
    void configure(const std::vector<size_t>& ends, const std::vector<int>& contributors, size_t minElems) {
        _ends = ends;
        _contributors = contributors;
        _minElems = minElems;
        beginStep();
    }
    size_t numTensors() const { return _ends.size(); }
    size_t totalElems() const { return _ends.empty() ? 0 : _ends.back(); }

    void beginStep() {
        _arrived.assign(_ends.size(), 0);
        _next = 0;
        _sent = _frontier = 0;
        advance();   // tensors nobody contributes to are never waited for
    }

    // One contribution to tensor `t` has been queued.  True when a bucket [*lo, *hi) is due now.
    bool arrived(size_t t, size_t* lo, size_t* hi) {
        if (t >= _ends.size()) return false;
        _arrived[t]++;
        advance();
        if (_frontier - _sent < _minElems) return false;
        *lo = _sent;
        *hi = _sent = _frontier;
        return true;
    }

    // End of back-propagation: the rest of the arena (true unless everything has been sent already).
    bool finish(size_t* lo, size_t* hi) {
        const size_t total = totalElems();
        if (_sent >= total) return false;
        *lo = _sent;
        *hi = _sent = _frontier = total;
        return true;
    }

private:
    void advance() {
        while (_next < _ends.size() && _arrived[_next] >= _contributors[_next]) _frontier = _ends[_next++];
    }
    std::vector<size_t> _ends;
    std::vector<int> _contributors, _arrived;
    size_t _minElems = 0, _next = 0, _sent = 0, _frontier = 0;
};

}  // namespace rt
