/*
 * ebvo_sort.h -- libstdc++'s std::sort, restated, on an index array with the two comparators of
 * Stereo_Matches::apply_Best_Nearly_Best_Test (src/Stereo_Matches.cpp:809-813):
 *     descending score:  comp(a, b) = score[a] > score[b]        (is_NCC)
 *     ascending score:   comp(a, b) = score[a] < score[b]
 * std::sort is not stable, and WHICH of two equal scores comes first decides which candidate survives at the cut of the
 * Best-Nearly-Best test and in what order the survivors continue through the chain.  The order is a property of the
 * algorithm libstdc++ implements (bits/stl_algo.h, bits/stl_heap.h; unchanged from GCC 4 to 14): introsort = quicksort
 * with the median of (first + 1, middle, last - 1) moved to the front as pivot and an unguarded Hoare partition,
 * recursing on the right part, down to ranges of 16, depth limit 2 * floor(log2(n)) then heapsort, and one final
 * insertion sort (guarded on the first 16, unguarded beyond).  Shared by the oracle and the kernels; checked against the
 * real std::sort by tests/test_oracle_glue.py (tests/cpp/sort_check.cpp).
 */
#ifndef EBVO_SORT_H
#define EBVO_SORT_H

#include <stdint.h>

#if defined(__HIPCC__)
#define EBVO_SORT_FN __device__ static inline
#else
#define EBVO_SORT_FN static inline
#endif

typedef struct
{
    const double *score;
    int descending;
} ebvo_sort_cmp;

EBVO_SORT_FN int ebvo_sort_less(const ebvo_sort_cmp *c, int32_t a, int32_t b)
{
    return c->descending ? c->score[a] > c->score[b] : c->score[a] < c->score[b];
}

EBVO_SORT_FN void ebvo_sort_swap(int32_t *a, int32_t *b)
{
    const int32_t t = *a;
    *a = *b;
    *b = t;
}

/* std::__unguarded_linear_insert */
EBVO_SORT_FN void ebvo_sort_unguarded_linear_insert(int32_t *v, int last, const ebvo_sort_cmp *c)
{
    const int32_t val = v[last];
    int next = last - 1;
    /* (next >= 0 never decides with a strict weak order; with NaN scores -- std::sort's behaviour is then undefined -- it
     * keeps the walk inside the row) */
    while (next >= 0 && ebvo_sort_less(c, val, v[next]))
    {
        v[last] = v[next];
        last = next;
        --next;
    }
    v[last] = val;
}

/* std::__insertion_sort on [first, last) */
EBVO_SORT_FN void ebvo_sort_insertion(int32_t *v, int first, int last, const ebvo_sort_cmp *c)
{
    if (first == last)
        return;
    for (int i = first + 1; i != last; ++i)
    {
        if (ebvo_sort_less(c, v[i], v[first]))
        {
            const int32_t val = v[i];
            for (int k = i; k > first; --k) /* std::move_backward(first, i, i + 1) */
                v[k] = v[k - 1];
            v[first] = val;
        }
        else
            ebvo_sort_unguarded_linear_insert(v, i, c);
    }
}

/* std::__push_heap / std::__adjust_heap on the heap v[first .. first + len) */
EBVO_SORT_FN void ebvo_sort_push_heap(int32_t *v, int first, int hole, int top, int32_t value, const ebvo_sort_cmp *c)
{
    int parent = (hole - 1) / 2;
    while (hole > top && ebvo_sort_less(c, v[first + parent], value))
    {
        v[first + hole] = v[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    v[first + hole] = value;
}

EBVO_SORT_FN void ebvo_sort_adjust_heap(int32_t *v, int first, int hole, int len, int32_t value, const ebvo_sort_cmp *c)
{
    const int top = hole;
    int second = hole;
    while (second < (len - 1) / 2)
    {
        second = 2 * (second + 1);
        if (ebvo_sort_less(c, v[first + second], v[first + second - 1]))
            second--;
        v[first + hole] = v[first + second];
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2)
    {
        second = 2 * (second + 1);
        v[first + hole] = v[first + second - 1];
        hole = second - 1;
    }
    ebvo_sort_push_heap(v, first, hole, top, value, c);
}

/* std::__partial_sort(first, last, last) = make_heap + sort_heap */
EBVO_SORT_FN void ebvo_sort_heapsort(int32_t *v, int first, int last, const ebvo_sort_cmp *c)
{
    const int len = last - first;
    if (len >= 2)
    {
        int parent = (len - 2) / 2;
        for (;;)
        {
            const int32_t value = v[first + parent];
            ebvo_sort_adjust_heap(v, first, parent, len, value, c);
            if (parent == 0)
                break;
            parent--;
        }
    }
    int l = last;
    while (l - first > 1)
    {
        --l;
        const int32_t value = v[l]; /* std::__pop_heap(first, l, l) */
        v[l] = v[first];
        ebvo_sort_adjust_heap(v, first, 0, l - first, value, c);
    }
}

/* std::sort(v, v + n, comp) */
EBVO_SORT_FN void ebvo_std_sort(int32_t *v, int n, const ebvo_sort_cmp *c)
{
    if (n < 2)
        return;
    int lg = 0;
    for (int t = n; t > 1; t >>= 1)
        ++lg;
    /* std::__introsort_loop with an explicit stack for the recursion on the right part */
    int st_first[64], st_last[64], st_depth[64], sp = 0;
    st_first[0] = 0;
    st_last[0] = n;
    st_depth[0] = lg * 2;
    sp = 1;
    while (sp > 0)
    {
        --sp;
        int first = st_first[sp], last = st_last[sp], depth = st_depth[sp];
        while (last - first > 16)
        {
            if (depth == 0)
            {
                ebvo_sort_heapsort(v, first, last, c);
                break;
            }
            --depth;
            /* std::__unguarded_partition_pivot */
            const int mid = first + (last - first) / 2;
            {
                const int a = first + 1, b = mid, cc = last - 1; /* std::__move_median_to_first(first, a, b, c) */
                if (ebvo_sort_less(c, v[a], v[b]))
                {
                    if (ebvo_sort_less(c, v[b], v[cc]))
                        ebvo_sort_swap(&v[first], &v[b]);
                    else if (ebvo_sort_less(c, v[a], v[cc]))
                        ebvo_sort_swap(&v[first], &v[cc]);
                    else
                        ebvo_sort_swap(&v[first], &v[a]);
                }
                else if (ebvo_sort_less(c, v[a], v[cc]))
                    ebvo_sort_swap(&v[first], &v[a]);
                else if (ebvo_sort_less(c, v[b], v[cc]))
                    ebvo_sort_swap(&v[first], &v[cc]);
                else
                    ebvo_sort_swap(&v[first], &v[b]);
            }
            int lo = first + 1, hi = last; /* std::__unguarded_partition(first + 1, last, first) */
            for (;;)
            {
                while (lo < last && ebvo_sort_less(c, v[lo], v[first])) /* bounds: see ebvo_sort_unguarded_linear_insert */
                    ++lo;
                --hi;
                while (hi > first && ebvo_sort_less(c, v[first], v[hi]))
                    --hi;
                if (!(lo < hi))
                    break;
                ebvo_sort_swap(&v[lo], &v[hi]);
                ++lo;
            }
            const int cut = lo;
            if (sp < 64) /* __introsort_loop(cut, last, depth_limit): the right part, with the depth reached here */
            {
                st_first[sp] = cut;
                st_last[sp] = last;
                st_depth[sp] = depth;
                ++sp;
            }
            last = cut;
        }
    }
    /* std::__final_insertion_sort */
    if (n > 16)
    {
        ebvo_sort_insertion(v, 0, 16, c);
        for (int i = 16; i != n; ++i)
            ebvo_sort_unguarded_linear_insert(v, i, c);
    }
    else
        ebvo_sort_insertion(v, 0, n, c);
}

#endif /* EBVO_SORT_H */
