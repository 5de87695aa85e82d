#include "../quokka_host.hpp"
