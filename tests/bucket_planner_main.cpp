// Drives rt::BucketPlanner (cuda-convnet_amd/csrc/runtime/buckets.h) from stdin for tests/test_dp_cpu.py:
//   line 1: minElems   line 2: tensor ends   line 3: contributors per tensor   line 4: arrival sequence (tensor indices)
// prints one "lo hi" line per bucket, in the order the runtime would launch the all-reduces (finish() last).
#include <iostream>
#include <sstream>
#include <string>

#include "buckets.h"

template <class T>
static std::vector<T> numbers(const std::string& line) {
    std::istringstream in(line);
    std::vector<T> v;
    for (T x; in >> x;) v.push_back(x);
    return v;
}

int main() {
    std::string l1, l2, l3, l4;
    std::getline(std::cin, l1);
    std::getline(std::cin, l2);
    std::getline(std::cin, l3);
    std::getline(std::cin, l4);
    rt::BucketPlanner planner;
    planner.configure(numbers<size_t>(l2), numbers<int>(l3), numbers<size_t>(l1).at(0));
    size_t lo, hi;
    for (int step = 0; step < 2; step++) {   // two steps: the second must plan exactly like the first
        planner.beginStep();
        for (size_t t : numbers<size_t>(l4))
            if (planner.arrived(t, &lo, &hi)) std::cout << lo << " " << hi << "\n";
        if (planner.finish(&lo, &hi)) std::cout << lo << " " << hi << "\n";
        std::cout << "end\n";
    }
    return 0;
}
