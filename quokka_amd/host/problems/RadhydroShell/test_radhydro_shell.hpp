#ifndef TEST_RADHYDRO_SHELL_HPP_
#define TEST_RADHYDRO_SHELL_HPP_
// function definitions
auto problem_main() -> int;
#endif
