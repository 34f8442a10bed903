// TEST INFRASTRUCTURE — a small stand-in for googletest (an empty submodule in the reference tree, not installed here): just
// enough of its interface for the reference's own pke unit tests (src/pke/unittest) to compile unmodified and run, so that they
// can be executed against the stock backend and against the HIP backend of DCRTPoly (tests/hal/Makefile, tests/test_ref_unittests.py).
// Supported: TEST, TEST_F, TEST_P, INSTANTIATE_TEST_SUITE_P (ValuesIn / Values, optional name generator), testing::Test,
// testing::TestWithParam<T>, TestParamInfo<T>, EXPECT_* / ASSERT_* {TRUE, FALSE, EQ, NE, LT, LE, GT, GE, NEAR, THROW, NO_THROW,
// ANY_THROW} with streamed messages, FAIL, SUCCEED, GTEST_SKIP, InitGoogleTest, RUN_ALL_TESTS with --gtest_filter=a:b-c and
// --gtest_list_tests.
#ifndef MINI_GTEST_H
#define MINI_GTEST_H

#include <cxxabi.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <type_traits>
#include <vector>

namespace testing {

class Test {
public:
    virtual ~Test() = default;
    virtual void SetUp() {}
    virtual void TearDown() {}
    virtual void TestBody() = 0;
};

// the result type helper functions of tests return (usable where a bool is expected, message ignored)
class AssertionResult {
public:
    explicit AssertionResult(bool ok) : ok_(ok) {}
    explicit operator bool() const {
        return ok_;
    }
    template <typename T>
    AssertionResult& operator<<(const T&) {
        return *this;
    }

private:
    bool ok_;
};
inline AssertionResult AssertionSuccess() {
    return AssertionResult(true);
}
inline AssertionResult AssertionFailure() {
    return AssertionResult(false);
}

template <typename T>
struct TestParamInfo {
    T param;
    size_t index;
};

template <typename T>
class WithParamInterface {
public:
    using ParamType = T;
    virtual ~WithParamInterface() = default;
    static const T& GetParam() {
        return *Current();
    }
    static const T*& Current() {
        static const T* p = nullptr;
        return p;
    }
};
template <typename T>
class TestWithParam : public Test, public WithParamInterface<T> {};

namespace internal {

struct TestEntry {
    std::string suite, name;
    std::function<void()> run;  // constructs the fixture, SetUp, TestBody, TearDown
};
struct State {
    std::vector<TestEntry> tests;
    std::vector<std::function<void()>> expanders;  // parameterised suites add their instances here
    bool currentFailed = false, currentSkipped = false;
    int failedAsserts = 0;
    std::string filter = "*";
    bool listOnly = false;
};
inline State& S() {
    static State s;
    return s;
}

inline bool Glob(const char* p, const char* s) {
    if (!*p)
        return !*s;
    if (*p == '*')
        return Glob(p + 1, s) || (*s && Glob(p, s + 1));
    if (*p == '?')
        return *s && Glob(p + 1, s + 1);
    return *p == *s && Glob(p + 1, s + 1);
}
inline bool AnyGlob(const std::string& patterns, const std::string& name) {
    size_t a = 0;
    while (a <= patterns.size()) {
        size_t b = patterns.find(':', a);
        if (b == std::string::npos)
            b = patterns.size();
        if (b > a && Glob(patterns.substr(a, b - a).c_str(), name.c_str()))
            return true;
        a = b + 1;
    }
    return false;
}
inline bool Selected(const std::string& full) {
    const std::string& f = S().filter;
    const size_t dash    = f.find('-');
    const std::string pos = dash == std::string::npos ? f : f.substr(0, dash);
    const std::string neg = dash == std::string::npos ? "" : f.substr(dash + 1);
    return AnyGlob(pos.empty() ? "*" : pos, full) && !(neg.size() && AnyGlob(neg, full));
}

template <typename Fixture>
void RunFixture() {
    Fixture f;
    ::testing::Test& t = f;  // (fixtures may declare their overrides protected / private, as googletest allows)
    t.SetUp();
    if (!S().currentSkipped)
        t.TestBody();
    t.TearDown();
}
inline int Register(const char* suite, const char* name, std::function<void()> run) {
    S().tests.push_back({suite, name, std::move(run)});
    return 0;
}

// one failed (or skipped) assertion: collects the streamed message, reports in the destructor
class Reporter {
public:
    Reporter(const char* file, int line, const std::string& what, bool fatalSkip = false) : skip_(fatalSkip) {
        os_ << file << ":" << line << ": " << what;
    }
    ~Reporter() {
        if (skip_) {
            S().currentSkipped = true;
            return;
        }
        S().currentFailed = true;
        ++S().failedAsserts;
        std::cout << os_.str() << std::endl;
    }
    template <typename T>
    Reporter& operator<<(const T& v) {
        os_ << v;
        return *this;
    }
    Reporter& operator<<(std::ostream& (*m)(std::ostream&)) {
        os_ << m;
        return *this;
    }

private:
    std::ostringstream os_;
    bool skip_;
};
// `if (ok) ; else Voidify() & Reporter(...) << msg` — the macro pattern that lets a message be streamed after the check
struct Voidify {
    void operator&(const Reporter&) const {}
};

template <typename T, typename = void>
struct Printable : std::false_type {};
template <typename T>
struct Printable<T, std::void_t<decltype(std::declval<std::ostream&>() << std::declval<const T&>())>> : std::true_type {};
template <typename T>
std::string Show(const T& v) {
    if constexpr (std::is_same_v<T, bool>)
        return v ? "true" : "false";
    else if constexpr (std::is_same_v<T, std::nullptr_t>)
        return "nullptr";
    else if constexpr (Printable<T>::value) {
        std::ostringstream os;
        os << v;
        return os.str();
    }
    else
        return "<object of " + std::to_string(sizeof(T)) + " bytes>";
}
template <typename A, typename B>
std::string CmpMsg(const char* ea, const char* eb, const char* op, const A& a, const B& b) {
    return std::string("Expected: (") + ea + ") " + op + " (" + eb + "), actual: " + Show(a) + " vs " + Show(b) + "\n";
}

// parameterised suites
template <typename Fixture>
struct ParamRegistry {
    using P = typename Fixture::ParamType;
    struct Pattern {
        std::string name;
        std::function<void()> body;  // RunFixture of the TEST_P class
    };
    static std::vector<Pattern>& Patterns() {
        static std::vector<Pattern> v;
        return v;
    }
    static int AddPattern(const char* name, std::function<void()> body) {
        Patterns().push_back({name, std::move(body)});
        return 0;
    }
    static int Instantiate(const char* prefix, const char* suite, std::vector<P> values, std::function<std::string(const TestParamInfo<P>&)> namer) {
        auto vals = std::make_shared<std::vector<P>>(std::move(values));
        S().expanders.push_back([=] {
            for (const auto& pat : Patterns())
                for (size_t i = 0; i < vals->size(); ++i) {
                    const std::string nm = namer ? namer(TestParamInfo<P>{(*vals)[i], i}) : std::to_string(i);
                    auto body            = pat.body;
                    S().tests.push_back({std::string(prefix) + "/" + suite, pat.name + "/" + nm, [vals, i, body] {
                                             WithParamInterface<P>::Current() = &(*vals)[i];
                                             body();
                                             WithParamInterface<P>::Current() = nullptr;
                                         }});
                }
        });
        return 0;
    }
};

template <typename C>
struct ValuesInGen {
    C values;
    template <typename P>
    std::vector<P> As() const {
        return std::vector<P>(std::begin(values), std::end(values));
    }
};
template <typename... Ts>
struct ValuesGen {
    std::tuple<Ts...> values;
    template <typename P>
    std::vector<P> As() const {
        std::vector<P> out;
        std::apply([&](const auto&... v) { (out.push_back(static_cast<P>(v)), ...); }, values);
        return out;
    }
};
struct NoNamer {};
template <typename P>
std::function<std::string(const TestParamInfo<P>&)> MakeNamer() {
    return nullptr;
}
template <typename P, typename F>
std::function<std::string(const TestParamInfo<P>&)> MakeNamer(F f) {
    return [f](const TestParamInfo<P>& i) { return std::string(f(i)); };
}

}  // namespace internal

template <typename C>
internal::ValuesInGen<C> ValuesIn(const C& c) {
    return {c};
}
template <typename T, size_t N>
internal::ValuesInGen<std::vector<T>> ValuesIn(const T (&a)[N]) {
    return {std::vector<T>(a, a + N)};
}
template <typename... Ts>
internal::ValuesGen<Ts...> Values(Ts... v) {
    return {std::make_tuple(v...)};
}

inline void InitGoogleTest(int* argc, char** argv) {
    int w = 1;
    for (int i = 1; i < *argc; ++i) {
        if (!std::strncmp(argv[i], "--gtest_filter=", 15))
            internal::S().filter = argv[i] + 15;
        else if (!std::strcmp(argv[i], "--gtest_list_tests"))
            internal::S().listOnly = true;
        else if (!std::strncmp(argv[i], "--gtest_", 8))
            ;
        else
            argv[w++] = argv[i];
    }
    *argc = w;
}
inline void InitGoogleTest() {}

}  // namespace testing

inline int RUN_ALL_TESTS() {
    using namespace testing::internal;
    for (auto& e : S().expanders)
        e();
    S().expanders.clear();
    int ran = 0, failed = 0, skipped = 0;
    std::vector<std::string> failedNames;
    const auto t0 = std::chrono::steady_clock::now();
    for (auto& t : S().tests) {
        const std::string full = t.suite + "." + t.name;
        if (!Selected(full))
            continue;
        if (S().listOnly) {
            std::cout << full << std::endl;
            continue;
        }
        S().currentFailed = S().currentSkipped = false;
        const auto a = std::chrono::steady_clock::now();
        try {
            t.run();
        }
        catch (const std::exception& e) {
            std::cout << "unexpected exception in " << full << ": " << e.what() << std::endl;
            S().currentFailed = true;
        }
        catch (...) {
            std::cout << "unexpected exception in " << full << std::endl;
            S().currentFailed = true;
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
        ++ran;
        if (S().currentSkipped)
            ++skipped;
        if (S().currentFailed) {
            ++failed;
            failedNames.push_back(full);
        }
        std::printf("[%s] %s (%.0f ms)\n", S().currentFailed ? "  FAILED  " : S().currentSkipped ? " SKIPPED  " : "       OK ", full.c_str(), ms);
        std::fflush(stdout);
    }
    if (S().listOnly)
        return 0;
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("[==========] %d tests ran, %d passed, %d failed, %d skipped (%.1f s)\n", ran, ran - failed - skipped, failed, skipped, sec);
    for (const auto& n : failedNames)
        std::printf("[  FAILED  ] %s\n", n.c_str());
    return failed ? 1 : 0;
}

#define MG_CAT_(a, b) a##b
#define MG_CAT(a, b) MG_CAT_(a, b)
#define MG_CLASS(suite, name) suite##_##name##_Test

#define TEST(suite, name)                                                                                                  \
    class MG_CLASS(suite, name) : public ::testing::Test {                                                                 \
        void TestBody() override;                                                                                          \
    };                                                                                                                     \
    static int MG_CAT(mg_reg_, __LINE__) = ::testing::internal::Register(#suite, #name, [] { ::testing::internal::RunFixture<MG_CLASS(suite, name)>(); }); \
    void MG_CLASS(suite, name)::TestBody()

#define TEST_F(fixture, name)                                                                                              \
    class MG_CLASS(fixture, name) : public fixture {                                                                       \
        void TestBody() override;                                                                                          \
    };                                                                                                                     \
    static int MG_CAT(mg_reg_, __LINE__) = ::testing::internal::Register(#fixture, #name, [] { ::testing::internal::RunFixture<MG_CLASS(fixture, name)>(); }); \
    void MG_CLASS(fixture, name)::TestBody()

#define TEST_P(fixture, name)                                                                                              \
    class MG_CLASS(fixture, name) : public fixture {                                                                       \
        void TestBody() override;                                                                                          \
    };                                                                                                                     \
    static int MG_CAT(mg_reg_, __LINE__) =                                                                                 \
        ::testing::internal::ParamRegistry<fixture>::AddPattern(#name, [] { ::testing::internal::RunFixture<MG_CLASS(fixture, name)>(); }); \
    void MG_CLASS(fixture, name)::TestBody()

#define INSTANTIATE_TEST_SUITE_P(prefix, fixture, generator, ...)                                                          \
    static int MG_CAT(mg_inst_, __LINE__) = ::testing::internal::ParamRegistry<fixture>::Instantiate(                      \
        #prefix, #fixture, (generator).template As<typename fixture::ParamType>(),                                        \
        ::testing::internal::MakeNamer<typename fixture::ParamType>(__VA_ARGS__))
#define INSTANTIATE_TEST_CASE_P INSTANTIATE_TEST_SUITE_P

#define MG_CHECK_(ok, text) \
    if (ok)                 \
        ;                   \
    else                    \
        ::testing::internal::Voidify() & ::testing::internal::Reporter(__FILE__, __LINE__, text)
#define MG_FATAL_(ok, text) \
    if (ok)                 \
        ;                   \
    else                    \
        return ::testing::internal::Voidify() & ::testing::internal::Reporter(__FILE__, __LINE__, text)

#define MG_CMP_(kind, a, b, op)                                                                          \
    if (const auto& mg_a = (a); true)                                                                    \
        if (const auto& mg_b = (b); true)                                                                \
    kind(mg_a op mg_b, ::testing::internal::CmpMsg(#a, #b, #op, mg_a, mg_b))

#define EXPECT_TRUE(c) MG_CHECK_(static_cast<bool>(c), std::string("Expected true: ") + #c + "\n")
#define EXPECT_FALSE(c) MG_CHECK_(!static_cast<bool>(c), std::string("Expected false: ") + #c + "\n")
#define ASSERT_TRUE(c) MG_FATAL_(static_cast<bool>(c), std::string("Expected true: ") + #c + "\n")
#define ASSERT_FALSE(c) MG_FATAL_(!static_cast<bool>(c), std::string("Expected false: ") + #c + "\n")
#define EXPECT_EQ(a, b) MG_CMP_(MG_CHECK_, a, b, ==)
#define EXPECT_NE(a, b) MG_CMP_(MG_CHECK_, a, b, !=)
#define EXPECT_LT(a, b) MG_CMP_(MG_CHECK_, a, b, <)
#define EXPECT_LE(a, b) MG_CMP_(MG_CHECK_, a, b, <=)
#define EXPECT_GT(a, b) MG_CMP_(MG_CHECK_, a, b, >)
#define EXPECT_GE(a, b) MG_CMP_(MG_CHECK_, a, b, >=)
#define ASSERT_EQ(a, b) MG_CMP_(MG_FATAL_, a, b, ==)
#define ASSERT_NE(a, b) MG_CMP_(MG_FATAL_, a, b, !=)
#define ASSERT_LT(a, b) MG_CMP_(MG_FATAL_, a, b, <)
#define ASSERT_LE(a, b) MG_CMP_(MG_FATAL_, a, b, <=)
#define ASSERT_GT(a, b) MG_CMP_(MG_FATAL_, a, b, >)
#define ASSERT_GE(a, b) MG_CMP_(MG_FATAL_, a, b, >=)
#define EXPECT_NEAR(a, b, eps) MG_CHECK_(std::fabs(double(a) - double(b)) <= double(eps), std::string("Expected near: ") + #a + " vs " + #b + "\n")
#define ASSERT_NEAR(a, b, eps) MG_FATAL_(std::fabs(double(a) - double(b)) <= double(eps), std::string("Expected near: ") + #a + " vs " + #b + "\n")
#define EXPECT_DOUBLE_EQ(a, b) EXPECT_NEAR(a, b, 4 * 2.220446049250313e-16 * std::fabs(double(a)))

#define MG_THROWS_(stmt, extype, flag) \
    bool flag = false;                 \
    try {                              \
        stmt;                          \
    }                                  \
    catch (const extype&) {            \
        flag = true;                   \
    }                                  \
    catch (...) {                      \
    }
#define EXPECT_THROW(stmt, extype)                         \
    if (bool mg_t = [&] { MG_THROWS_(stmt, extype, f) return f; }(); true) \
    MG_CHECK_(mg_t, std::string("Expected exception ") + #extype + " from " + #stmt + "\n")
#define EXPECT_ANY_THROW(stmt)                                                                              \
    if (bool mg_t = [&] { bool f = false; try { stmt; } catch (...) { f = true; } return f; }(); true) \
    MG_CHECK_(mg_t, std::string("Expected an exception from ") + #stmt + "\n")
#define EXPECT_NO_THROW(stmt)                                                                               \
    if (bool mg_t = [&] { bool f = true; try { stmt; } catch (...) { f = false; } return f; }(); true) \
    MG_CHECK_(mg_t, std::string("Unexpected exception from ") + #stmt + "\n")
#define ASSERT_THROW EXPECT_THROW
#define ASSERT_ANY_THROW EXPECT_ANY_THROW
#define ASSERT_NO_THROW EXPECT_NO_THROW

#define FAIL() return ::testing::internal::Voidify() & ::testing::internal::Reporter(__FILE__, __LINE__, "Failed\n")
#define ADD_FAILURE() ::testing::internal::Voidify() & ::testing::internal::Reporter(__FILE__, __LINE__, "Failed\n")
#define SUCCEED() \
    if (true)     \
        ;         \
    else          \
        ::testing::internal::Voidify() & ::testing::internal::Reporter(__FILE__, __LINE__, "")
#define GTEST_SKIP() return ::testing::internal::Voidify() & ::testing::internal::Reporter(__FILE__, __LINE__, "skipped", true)

#endif
