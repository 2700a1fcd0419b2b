// Set-associative LRU replay for profiles/l2_sweep_sim.py: argv = stream file (u64 line numbers), capacity in lines, ways, hash.
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>
int main(int argc, char** argv) {
    const char* path = argv[1];
    const int64_t cap_lines = atoll(argv[2]);   // e.g. 32768 for 4 MiB
    const int ways = atoi(argv[3]);
    const int hash = argc > 4 ? atoi(argv[4]) : 0;
    const int64_t nsets = cap_lines / ways;
    std::vector<uint64_t> tag(cap_lines, ~0ull);
    std::vector<uint32_t> age(cap_lines, 0);
    FILE* f = fopen(path, "rb");
    std::vector<uint64_t> buf(1 << 20);
    uint64_t miss = 0, total = 0; uint32_t clock = 0;
    size_t n;
    while ((n = fread(buf.data(), 8, buf.size(), f)) > 0) {
        for (size_t i = 0; i < n; i++) {
            uint64_t l = buf[i];
            uint64_t s = l;
            if (hash) s = l ^ (l >> 11) ^ (l >> 22);
            s %= nsets;
            uint64_t* t = &tag[s * ways]; uint32_t* a = &age[s * ways];
            int hit = -1, victim = 0; uint32_t oldest = 0xffffffffu;
            for (int w = 0; w < ways; w++) {
                if (t[w] == l) { hit = w; break; }
                if (t[w] == ~0ull) { victim = w; oldest = 0; }
                else if (oldest != 0 && a[w] < oldest) { oldest = a[w]; victim = w; }
            }
            ++clock; ++total;
            if (hit >= 0) a[hit] = clock; else { ++miss; t[victim] = l; a[victim] = clock; }
        }
    }
    printf("total %llu miss %llu\n", (unsigned long long)total, (unsigned long long)miss);
}
