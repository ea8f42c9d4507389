// tests/duckdb_stub: forwards to the one stub header (see duckdb.hpp in this directory tree)
#pragma once
#include "duckdb.hpp"
