/* scripts/callers_loop.c — T native threads looping `usearch_search` (one query per call), for scripts/coalesce_check.py: Python
 * threads hand the interpreter lock around between calls, which is not what a Go or C# caller of the C ABI does. */
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <time.h>

typedef size_t (*search_t)(void*, void const*, int, size_t, uint64_t*, float*, char const**);

typedef struct {
    search_t search;
    void* index;
    char const* queries;
    size_t query_bytes, queries_count, wanted, first;
    int kind, calls, failures;
} caller_t;

static void* caller_loop(void* raw) {
    caller_t* caller = (caller_t*)raw;
    uint64_t* keys = (uint64_t*)malloc(caller->wanted * 8);
    float* distances = (float*)malloc(caller->wanted * 4);
    for (int i = 0; i < caller->calls; ++i) {
        char const* error = NULL;
        caller->search(caller->index, caller->queries + ((caller->first + i) % caller->queries_count) * caller->query_bytes, caller->kind,
                       caller->wanted, keys, distances, &error);
        if (error)
            ++caller->failures;
    }
    free(keys), free(distances);
    return NULL;
}

/* Seconds the T threads took for `calls` searches each; `failures` gets the number of calls that reported an error. */
double callers_loop(search_t search, void* index, void const* queries, size_t query_bytes, size_t queries_count, int kind, size_t wanted,
                    int threads, int calls, int* failures) {
    caller_t* callers = (caller_t*)calloc((size_t)threads, sizeof(caller_t));
    pthread_t* pool = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    struct timespec begin, end;
    clock_gettime(CLOCK_MONOTONIC, &begin);
    for (int t = 0; t < threads; ++t) {
        caller_t c = {search, index, (char const*)queries, query_bytes, queries_count, wanted, (size_t)t * (size_t)calls, kind, calls, 0};
        callers[t] = c;
        pthread_create(&pool[t], NULL, caller_loop, &callers[t]);
    }
    *failures = 0;
    for (int t = 0; t < threads; ++t)
        pthread_join(pool[t], NULL), *failures += callers[t].failures;
    clock_gettime(CLOCK_MONOTONIC, &end);
    free(callers), free(pool);
    return (double)(end.tv_sec - begin.tv_sec) + 1e-9 * (double)(end.tv_nsec - begin.tv_nsec);
}
