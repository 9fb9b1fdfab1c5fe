// gtest.h -- the few lines of the GoogleTest surface that the reference's own test programs
// (test/test_{bfv,ckks,tfhe}_*.cpp: TEST and EXPECT_EQ only) use, so that they can be compiled
// UNCHANGED from where they lie and run against this backend's class layer
// (include/heongpu/heongpu.hpp).  GoogleTest itself is not in the image.  Test infrastructure.
// The reference's tests define main() themselves (InitGoogleTest + RUN_ALL_TESTS).
#pragma once
#include <cstdio>
#include <exception>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

namespace testing {
struct Registry {
    struct Case { const char* suite; const char* name; void (*fn)(); };
    static std::vector<Case>& cases() { static std::vector<Case> c; return c; }
    static int& failures() { static int f = 0; return f; }
    static int add(const char* suite, const char* name, void (*fn)()) { cases().push_back({suite, name, fn}); return 0; }
};
// `EXPECT_EQ(a, b) << "message"` is legal GoogleTest; the reference does not stream, but keep it valid
struct Sink {
    template <typename T> Sink& operator<<(const T&) { return *this; }
};
template <typename A, typename B> bool expect_eq(const A& a, const B& b, const char* ea, const char* eb, const char* file, int line)
{
    if (a == b) return true;
    Registry::failures()++;
    std::cout << file << ":" << line << ": Failure\nExpected equality of these values:\n  " << ea << "\n  " << eb << std::endl;
    return false;
}
inline void InitGoogleTest(int*, char**) {}
} // namespace testing

#define TEST(suite, name)                                                                              \
    static void suite##_##name##_body();                                                               \
    static int suite##_##name##_registered = ::testing::Registry::add(#suite, #name, suite##_##name##_body); \
    static void suite##_##name##_body()
#define EXPECT_EQ(a, b) (::testing::expect_eq((a), (b), #a, #b, __FILE__, __LINE__), ::testing::Sink())
#define EXPECT_TRUE(a) EXPECT_EQ((bool) (a), true)
#define EXPECT_FALSE(a) EXPECT_EQ((bool) (a), false)

inline int RUN_ALL_TESTS()
{
    int failed_cases = 0;
    for (const auto& c : ::testing::Registry::cases()) {
        std::cout << "[ RUN      ] " << c.suite << "." << c.name << std::endl;
        const int before = ::testing::Registry::failures();
        try {
            c.fn();
        } catch (const std::exception& e) {
            ::testing::Registry::failures()++;
            std::cout << "unexpected exception: " << e.what() << std::endl;
        }
        const bool ok = ::testing::Registry::failures() == before;
        failed_cases += !ok;
        std::cout << (ok ? "[       OK ] " : "[  FAILED  ] ") << c.suite << "." << c.name << std::endl;
    }
    std::cout << "[==========] " << ::testing::Registry::cases().size() << " tests ran, " << failed_cases << " failed." << std::endl;
    return failed_cases ? 1 : 0;
}
