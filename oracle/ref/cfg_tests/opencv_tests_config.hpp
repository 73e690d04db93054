/* what cmake would generate for the reference's test binaries (modules/ts/src/ts.cpp:109): no installed test data in the oracle build */
