// acr::BIH<T, MAX_DEPTH> -- bounding interval hierarchy, host build.
//
// Same public shape as the reference's template (src/utils/bih.h:68-89): a
// complete binary tree in heap order (children of i at 2i+1 and 2i+2), MAX_SIZE
// = 2^MAX_DEPTH - 1 nodes, built once in the constructor by sift(), which
// permutes the item array in place.  Differences on this side of the boundary:
//   * traversal is not here: it runs on the GPU (device/acr_kernels.hip);
//   * the depth may also be chosen at run time (constructor argument), because
//     one of the workloads needs a far deeper tree than the reference's 6;
//     MAX_DEPTH stays the default;
//   * nodes are zero-initialised and carry a `defined` mask saying which
//     fields sift() wrote, since the reference leaves the others
//     indeterminate (bih.h:139-143);
//   * `order` records where each item came from, so a hit can be reported as
//     the caller's original primitive number.
#ifndef ACR_UTILS_BIH_H
#define ACR_UTILS_BIH_H

#include <cstddef>
#include <cstdint>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "../geometry/geometry.h"
#include "../math/math.h"
#include "vector.h"

namespace acr
{
	template <typename T, size_t MAX_DEPTH = 6>
	class BIH
	{
	public:
		struct Node // bih.h:92-98
		{
			uint32_t start = 0, end = 0;
			float left = 0, right = 0;
			uint8_t axis = 0;
			uint8_t isLeaf = 0;
			uint8_t defined = 0; // bit0 start/end, bit1 left/right/axis, bit2 isLeaf
		};

		BIH() : depth(MAX_DEPTH) {}
		BIH(const BIH &) = default;

		// bih.h:75-81
		BIH(std::vector<T> &hObjs, const BoundingBox &bb, void *data, size_t maxDepth = MAX_DEPTH)
			: boundingBox(bb), depth(maxDepth), tree((size_t(1) << maxDepth) - 1)
		{
			std::vector<uint32_t> ord(hObjs.size());
			for (size_t i = 0; i < ord.size(); i++) ord[i] = uint32_t(i);
			Build b{this, hObjs, ord, data};
			// The two recursive calls of sift() work on disjoint item ranges and disjoint subtrees, so large
			// builds run them as parallel tasks; every node sees the same items in the same order as in a
			// serial build, so the tree and the permutation are identical.
#pragma omp parallel if (hObjs.size() >= kParallelBuildItems)
#pragma omp single nowait
			b.sift(0, boundingBox, 0, hObjs.size());
			objs = vector<T>(hObjs);
			order = ord;
		}

		const static size_t MAX_SIZE = (size_t(1) << MAX_DEPTH) - 1; // bih.h:89, for the default depth
		const static size_t kParallelBuildItems = 1 << 15;           // subtrees with fewer items are built serially
		size_t maxSize() const { return tree.size(); }
		size_t maxDepth() const { return depth; }
		const std::vector<Node> &nodes() const { return tree; }
		const vector<T> &items() const { return objs; }
		const std::vector<uint32_t> &itemOrder() const { return order; }
		const BoundingBox &bounds() const { return boundingBox; }

		// the printout of bih.h:301-314
		std::string toString() const
		{
			std::ostringstream os;
			if (!tree.empty()) outputNode(os, 0, 0);
			return os.str();
		}

	private:
		struct Build
		{
			BIH *self;
			std::vector<T> &h;
			std::vector<uint32_t> &ord;
			void *data;

			// bih.h:129-207.  Returns true when `index` is outside the heap or holds no items.
			bool sift(size_t index, const BoundingBox &bb, size_t start, size_t end)
			{
				if (index >= self->tree.size())
					return true;
				Node &node = self->tree[index];
				node.start = uint32_t(start);
				node.end = uint32_t(end);
				node.defined |= 1;
				if (start >= end) {
					node.isLeaf = 1;
					node.defined |= 4;
					return true;
				}

				// longest axis of the box handed down, first one on ties (bih.h:146-156)
				float len = -1;
				uint8_t axis = 0;
				for (int k = 0; k < 3; k++) {
					const float axisLen = bb.max[k] - bb.min[k];
					if (axisLen > len) {
						len = axisLen;
						axis = uint8_t(k);
					}
				}
				const float pivot = bb.min[axis] + len / 2;

				// child boxes start inverted and grow over their members (bih.h:162-168)
				BoundingBox lBB;
				lBB.min = bb.max;
				lBB.max = bb.min;
				BoundingBox rBB = lBB;

				// one pass: items left of the pivot stay, others are swapped to the back (bih.h:170-195)
				long lo = long(start), hi = long(end) - 1;
				while (lo <= hi) {
					const BoundingBox ibb = h[lo].getBoundingBox(data);
					const float pos = h[lo].getCentroid(data)[axis];
					if (pos <= pivot) {
						lBB.min = math::min(lBB.min, ibb.min);
						lBB.max = math::max(lBB.max, ibb.max);
						lo++;
					} else {
						rBB.min = math::min(rBB.min, ibb.min);
						rBB.max = math::max(rBB.max, ibb.max);
						if (lo != hi) {
							std::swap(h[lo], h[hi]);
							std::swap(ord[lo], ord[hi]);
						}
						hi--;
					}
				}

				node.left = lBB.max[axis];
				node.right = rBB.min[axis];
				node.axis = axis;
				node.defined |= 2;

				bool left, right;
				if (end - start >= kParallelBuildItems) {
#pragma omp task shared(left) firstprivate(index, lBB, start, lo)
					left = sift(2 * index + 1, lBB, start, size_t(lo));
					right = sift(2 * index + 2, rBB, size_t(lo), end);
#pragma omp taskwait
				} else {
					left = sift(2 * index + 1, lBB, start, size_t(lo));
					right = sift(2 * index + 2, rBB, size_t(lo), end);
				}
				// a node is a leaf as soon as EITHER child is missing or empty (bih.h:203-205)
				self->tree[index].isLeaf = uint8_t(left || right);
				self->tree[index].defined |= 4;
				return false;
			}
		};

		void outputNode(std::ostream &os, size_t index, size_t level) const
		{
			os << std::string(level, '\t') << index << ": [" << tree[index].start << ":" << tree[index].end << "]" << std::endl;
			if (!tree[index].isLeaf) {
				outputNode(os, 2 * index + 1, level + 1);
				outputNode(os, 2 * index + 2, level + 1);
			}
		}

		BoundingBox boundingBox; // bih.h:316
		size_t depth;
		std::vector<Node> tree;  // bih.h:317
		vector<T> objs;          // bih.h:318
		std::vector<uint32_t> order;
	};
} // namespace acr

#endif
