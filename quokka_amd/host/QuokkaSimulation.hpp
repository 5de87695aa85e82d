#include "quokka_amr.hpp"
