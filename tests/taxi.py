"""Loads the reference's taxi sample (tests/golden/taxi_sample_header.csv = the data file of
omniscidb/Tests/ArrowStorageSqlTest.cpp / python/tests/test_pyhdk_api.py) with the column types of
omniscidb/Benchmarks/taxi/taxi_reduced_bench.cpp:13-24, and defines Q1-Q4 as QueryUnits."""
import os

import pyarrow as pa
import pyarrow.csv as pcsv

from hdk_amd.ir import Agg, Cast, ColRef, ExtractYear, INT32, KeyRef, QueryUnit

HERE = os.path.dirname(os.path.abspath(__file__))


def load_taxi(storage, fragment_size=None):
    conv = pcsv.ConvertOptions(column_types={
        "passenger_count": pa.int16(), "pickup_datetime": pa.timestamp("s"),
        "trip_distance": pa.decimal128(14, 2), "total_amount": pa.decimal128(14, 2), "cab_type": pa.string()},
        include_columns=["pickup_datetime", "passenger_count", "trip_distance", "total_amount", "cab_type"])
    at = pcsv.read_csv(os.path.join(HERE, "golden", "taxi_sample_header.csv"), convert_options=conv)
    return storage.import_arrow(at, "trips", fragment_size=fragment_size)


def taxi_queries():
    q1 = QueryUnit("trips", groupby=[ColRef("cab_type")],
                   targets=[KeyRef(0, "cab_type"), Agg("count", None, "cnt")])
    q2 = QueryUnit("trips", groupby=[ColRef("passenger_count")],
                   targets=[KeyRef(0, "passenger_count"), Agg("avg", ColRef("total_amount"), "total_amount_avg")])
    q3 = QueryUnit("trips", groupby=[ColRef("passenger_count"), ExtractYear(ColRef("pickup_datetime"))],
                   targets=[KeyRef(0, "passenger_count"), KeyRef(1, "pickup_year"), Agg("count", None, "cnt")])
    q4 = QueryUnit("trips", groupby=[ColRef("passenger_count"), ExtractYear(ColRef("pickup_datetime")),
                                     Cast(ColRef("trip_distance"), INT32)],
                   targets=[KeyRef(0, "passenger_count"), KeyRef(1, "pickup_year"), KeyRef(2, "distance"),
                            Agg("count", None, "cnt")])
    return q1, q2, q3, q4


def check_taxi_results(cols1, cols2, cols3, cols4):
    """Expected values: python/tests/test_pyhdk_api.py:1322-1358 / ArrowStorageSqlTest.cpp:196-244."""
    assert cols1 == {"cab_type": ["green"], "cnt": [20]}
    o = sorted(range(len(cols2["passenger_count"])), key=lambda i: cols2["passenger_count"][i])
    assert [cols2["passenger_count"][i] for i in o] == [1, 2, 5]
    got = [cols2["total_amount_avg"][i] for i in o]
    want = [98.19 / 16, 75.0, 13.58 / 3]
    assert all(abs(g - w) <= 1e-9 * max(1.0, abs(w)) for g, w in zip(got, want)), got
    o = sorted(range(len(cols3["passenger_count"])), key=lambda i: cols3["passenger_count"][i])
    assert [cols3["passenger_count"][i] for i in o] == [1, 2, 5]
    assert [cols3["pickup_year"][i] for i in o] == [2013, 2013, 2013]
    assert [cols3["cnt"][i] for i in o] == [16, 1, 3]
    o = sorted(range(len(cols4["cnt"])), key=lambda i: (cols4["pickup_year"][i], -cols4["cnt"][i]))
    assert [cols4["passenger_count"][i] for i in o] == [1, 5, 2]
    assert [cols4["pickup_year"][i] for i in o] == [2013, 2013, 2013]
    assert [cols4["distance"][i] for i in o] == [0, 0, 0]
    assert [cols4["cnt"][i] for i in o] == [16, 3, 1]
